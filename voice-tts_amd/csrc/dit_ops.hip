// dit_ops.hip -- fused row / element kernels of the s2mel DiT (row N1), fp32 as the reference runs this stage
// (infer_v2.py:710-711).  Each replaces a chain of 2-6 framework kernels between the GEMMs; all are HBM-bound
// single passes (8 B per element read+written) except where noted.
//
//   adaln_rmsnorm   AdaptiveLayerNorm(RMSNorm): bias + weight * (rms_norm(x) * g)      gpt_fast/model.py:18-37,362-372
//   ln_modulate     FinalLayer: layer_norm(x, eps 1e-6, no affine) * (1+scale) + shift  diffusion_transformer.py:83-100
//   rope_qk         apply_rotary_emb on the q and k thirds of the wqkv output, in place   gpt_fast/model.py:289-301,348-360
//   swiglu          silu(w1 x) * (w3 x) on the fused [w1; w3] GEMM output                gpt_fast/model.py:316-326
//   wn_gate         tanh(x_in + g_a) * sigmoid(x_in + g_b) (fused_add_tanh_sigmoid_multiply)  wavenet.py:142-160, commons.py
#include "common.h"

namespace ixtts {

constexpr int ROW_NV = 8;  // float4 per lane: rows up to 64 * 8 * 4 = 2048 floats

// one wavefront per row; the row lives in registers between the statistics and the output pass
template <bool CENTER>
__global__ __launch_bounds__(256) void row_norm_mod_kernel(const float* __restrict__ x, const float* __restrict__ mw, const float* __restrict__ mb,
                                                           const float* __restrict__ g, float* __restrict__ out, long rows, int T, int H, long mstride,
                                                           float eps, float wadd) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int H4 = H >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + row * H);
  float4 v[ROW_NV];
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < H4) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float mean = 0.f;
  if constexpr (CENTER) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    mean = wave_sum(s) / (float)H;
  }
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int c = lane + 64 * i;
    if (c < H4) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
  const long bidx = row / T;  // modulation vectors are per batch entry
  const float4* w4 = reinterpret_cast<const float4*>(mw + bidx * mstride);
  const float4* b4 = reinterpret_cast<const float4*>(mb + bidx * mstride);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* o4 = reinterpret_cast<float4*>(out + row * H);
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int c = lane + 64 * i;
    if (c < H4) {
      const float4 w = w4[c], b = b4[c];
      float4 n = make_float4((v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd);
      if (g) {
        const float4 gg = g4[c];
        n.x *= gg.x; n.y *= gg.y; n.z *= gg.z; n.w *= gg.w;
      }
      o4[c] = make_float4(fmaf(w.x + wadd, n.x, b.x), fmaf(w.y + wadd, n.y, b.y), fmaf(w.z + wadd, n.z, b.z), fmaf(w.w + wadd, n.w, b.w));
    }
  }
}

// qkv [rows][3H]; q = cols [0,H), k = cols [H,2H); pair i of head h sits at h*hd + 2i, +1; table [T][hd/2] (cos, sin)
__global__ __launch_bounds__(256) void rope_qk_kernel(float* __restrict__ qkv, const float2* __restrict__ tab, long rows, int T, int H, int hd) {
  const int per_row = (2 * H) >> 2;  // float4 (two pairs) of the q and k thirds
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * per_row) return;
  const long row = idx / per_row;
  const int c4 = (int)(idx % per_row);
  const int col = c4 * 4;  // within [0, 2H)
  const int t = (int)(row % T);
  const int p = (col % hd) >> 1;  // first of the two pairs
  float4* ptr = reinterpret_cast<float4*>(qkv + row * 3 * H + col);
  const float4 v = *ptr;
  const float2 f0 = tab[(long)t * (hd >> 1) + p], f1 = tab[(long)t * (hd >> 1) + p + 1];
  *ptr = make_float4(v.x * f0.x - v.y * f0.y, v.y * f0.x + v.x * f0.y, v.z * f1.x - v.w * f1.y, v.w * f1.x + v.z * f1.y);
}

__global__ __launch_bounds__(256) void swiglu_kernel(const float* __restrict__ u, float* __restrict__ out, long rows, int Fd) {
  const int F4 = Fd >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * F4) return;
  const long row = idx / F4;
  const int c = (int)(idx % F4);
  const float4 a = reinterpret_cast<const float4*>(u + row * 2 * Fd)[c];
  const float4 b = reinterpret_cast<const float4*>(u + row * 2 * Fd + Fd)[c];
  auto f = [](float x, float y) { return x / (1.0f + expf(-x)) * y; };
  reinterpret_cast<float4*>(out + row * Fd)[c] = make_float4(f(a.x, b.x), f(a.y, b.y), f(a.z, b.z), f(a.w, b.w));
}

// a [B][2C][T]; gvec [B][gstride] with the layer's 2C gate biases at goff; out [B][C][T]
__global__ __launch_bounds__(256) void wn_gate_kernel(const float* __restrict__ a, const float* __restrict__ gvec, float* __restrict__ out, int B, int C, int T,
                                                      long gstride, int goff) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = (long)B * C * T;
  if (idx >= n) return;
  const int t = (int)(idx % T);
  const int c = (int)((idx / T) % C);
  const int b = (int)(idx / ((long)T * C));
  const float ga = gvec[b * gstride + goff + c], gb = gvec[b * gstride + goff + C + c];
  const float xa = a[((long)b * 2 * C + c) * T + t] + ga;
  const float xb = a[((long)b * 2 * C + C + c) * T + t] + gb;
  out[idx] = tanhf(xa) * (1.0f / (1.0f + expf(-xb)));
}

// Row layout of the same gate: a [rows][2C], out [rows][C]; row r belongs to batch entry min(r / rows_per_batch, B-1)
__global__ __launch_bounds__(256) void wn_gate_rows_kernel(const float* __restrict__ a, const float* __restrict__ gvec, float* __restrict__ out, long rows, int C,
                                                           long rows_per_batch, int B, long gstride, int goff) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // one float4 of one output row
  const int C4 = C >> 2;
  if (idx >= rows * C4) return;
  const long r = idx / C4;
  const int c = (int)(idx % C4) * 4;
  const int b = (int)min(r / rows_per_batch, (long)B - 1);
  const float4 xa = *reinterpret_cast<const float4*>(a + r * 2 * C + c);
  const float4 xb = *reinterpret_cast<const float4*>(a + r * 2 * C + C + c);
  const float4 ga = *reinterpret_cast<const float4*>(gvec + b * gstride + goff + c);
  const float4 gb = *reinterpret_cast<const float4*>(gvec + b * gstride + goff + C + c);
  float4 o;
  o.x = tanhf(xa.x + ga.x) * (1.0f / (1.0f + expf(-(xb.x + gb.x))));
  o.y = tanhf(xa.y + ga.y) * (1.0f / (1.0f + expf(-(xb.y + gb.y))));
  o.z = tanhf(xa.z + ga.z) * (1.0f / (1.0f + expf(-(xb.z + gb.z))));
  o.w = tanhf(xa.w + ga.w) * (1.0f / (1.0f + expf(-(xb.w + gb.w))));
  *reinterpret_cast<float4*>(out + r * C + c) = o;
}

// Reflect halo of a row-layout sequence buffer p [B][left + T + right][C]: halo row left-1-i = interior row i+1,
// halo row left+T+i = interior row T-2-i (torch's reflect padding, wavenet.py's SConv1d pad_mode)
__global__ __launch_bounds__(256) void reflect_halo_rows_kernel(float* __restrict__ p, int B, int T, int C, int left, int right) {
  const int C4 = C >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long per_b = (long)(left + right) * C4;
  if (idx >= per_b * B) return;
  const int b = (int)(idx / per_b);
  const int h = (int)((idx % per_b) / C4);
  const int c = (int)(idx % C4) * 4;
  float* base = p + (long)b * (left + T + right) * C;
  const int dst = h < left ? left - 1 - h : left + T + (h - left);
  const int src = h < left ? left + h + 1 : left + T - 2 - (h - left);
  *reinterpret_cast<float4*>(base + (long)dst * C + c) = *reinterpret_cast<const float4*>(base + (long)src * C + c);
}

}  // namespace ixtts

using namespace ixtts;

extern "C" int ixtts_adaln_rmsnorm_f32(const float* x_dev, const float* wb_dev, const float* g_dev, float* out_dev, int B, int T, int H, float eps,
                                       void* stream) {
  IX_ARG(x_dev && wb_dev && g_dev && out_dev, "adaln_rmsnorm: null pointer");
  IX_ARG(B > 0 && T > 0 && H > 0 && H % 4 == 0 && H <= 64 * ROW_NV * 4, "adaln_rmsnorm: bad shape B=%d T=%d H=%d", B, T, H);
  const long rows = (long)B * T;
  hipLaunchKernelGGL(row_norm_mod_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x_dev, wb_dev, wb_dev + H, g_dev, out_dev,
                     rows, T, H, (long)2 * H, eps, 0.0f);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

extern "C" int ixtts_ln_modulate_f32(const float* x_dev, const float* shift_scale_dev, float* out_dev, int B, int T, int H, float eps, void* stream) {
  IX_ARG(x_dev && shift_scale_dev && out_dev, "ln_modulate: null pointer");
  IX_ARG(B > 0 && T > 0 && H > 0 && H % 4 == 0 && H <= 64 * ROW_NV * 4, "ln_modulate: bad shape B=%d T=%d H=%d", B, T, H);
  const long rows = (long)B * T;
  // shift_scale [B][2H] = (shift | scale): out = norm * (1 + scale) + shift
  hipLaunchKernelGGL(row_norm_mod_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x_dev, shift_scale_dev + H, shift_scale_dev,
                     (const float*)nullptr, out_dev, rows, T, H, (long)2 * H, eps, 1.0f);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

extern "C" int ixtts_rope_qk_f32(float* qkv_dev, const float* cos_sin_dev, int B, int T, int H, int head_dim, void* stream) {
  IX_ARG(qkv_dev && cos_sin_dev, "rope_qk: null pointer");
  IX_ARG(B > 0 && T > 0 && H > 0 && head_dim > 0 && head_dim % 4 == 0 && H % head_dim == 0, "rope_qk: bad shape B=%d T=%d H=%d hd=%d", B, T, H, head_dim);
  const long n = (long)B * T * (2 * H / 4);
  hipLaunchKernelGGL(rope_qk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qkv_dev, reinterpret_cast<const float2*>(cos_sin_dev),
                     (long)B * T, T, H, head_dim);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

extern "C" int ixtts_swiglu_f32(const float* u_dev, float* out_dev, long rows, int F, void* stream) {
  IX_ARG(u_dev && out_dev, "swiglu: null pointer");
  IX_ARG(rows > 0 && F > 0 && F % 4 == 0, "swiglu: bad shape rows=%ld F=%d", rows, F);
  const long n = rows * (F / 4);
  hipLaunchKernelGGL(swiglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u_dev, out_dev, rows, F);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

extern "C" int ixtts_wn_gate_f32(const float* a_dev, const float* g_dev, float* out_dev, int B, int C, int T, long g_stride, int g_offset, void* stream) {
  IX_ARG(a_dev && g_dev && out_dev, "wn_gate: null pointer");
  IX_ARG(B > 0 && C > 0 && T > 0 && g_offset >= 0, "wn_gate: bad shape B=%d C=%d T=%d", B, C, T);
  const long n = (long)B * C * T;
  hipLaunchKernelGGL(wn_gate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a_dev, g_dev, out_dev, B, C, T, g_stride, g_offset);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

extern "C" int ixtts_wn_gate_rows_f32(const float* a_dev, const float* g_dev, float* out_dev, long rows, int C, long rows_per_batch, int B, long g_stride,
                                      int g_offset, void* stream) {
  IX_ARG(a_dev && g_dev && out_dev, "wn_gate_rows: null pointer");
  IX_ARG(rows > 0 && C > 0 && C % 4 == 0 && rows_per_batch > 0 && B > 0 && g_offset >= 0 && g_offset % 4 == 0 && g_stride % 4 == 0,
         "wn_gate_rows: bad shape rows=%ld C=%d rows_per_batch=%ld B=%d", rows, C, rows_per_batch, B);
  const long n = rows * (C / 4);
  hipLaunchKernelGGL(wn_gate_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a_dev, g_dev, out_dev, rows, C, rows_per_batch, B,
                     g_stride, g_offset);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

extern "C" int ixtts_reflect_halo_rows_f32(float* p_dev, int B, int T, int C, int left, int right, void* stream) {
  IX_ARG(p_dev, "reflect_halo_rows: null pointer");
  IX_ARG(B > 0 && C > 0 && C % 4 == 0 && left >= 0 && right >= 0 && T > left && T > right, "reflect_halo_rows: bad shape B=%d T=%d C=%d left=%d right=%d", B, T, C,
         left, right);
  const long n = (long)B * (left + right) * (C / 4);
  if (n == 0) return IXTTS_OK;
  hipLaunchKernelGGL(reflect_halo_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p_dev, B, T, C, left, right);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}
