// attn_full.hip -- full (non-causal, unmasked) multi-head attention in exact fp32 on the fp32 matrix cores of
// gfx950, for the s2mel DiT (row N1): 25 Euler steps x 13 layers x (batch 2 x 8 heads) x T ~ 2300 x d = 64.
//
// Reference op: F.scaled_dot_product_attention(q, k, v, attn_mask=all-true) in
// indextts/s2mel/modules/gpt_fast/model.py:303 (fp32: the reference runs s2mel without autocast, infer_v2.py:710-711).
//
// Flash-style, no S x S matrix, and NO transposes / LDS round trip between the two products:
//   S^T = K Q^T   (v_mfma_f32_32x32x2_f32; rows = keys, cols = queries) leaves, in accumulator register r of lane l,
//                 the score of query (l & 31) against key  (r&3) + 8*(r>>2) + 4*(l>>5).
//   O^T += V^T P^T  sums over keys; the MFMA B operand of k-step r is B[k = l>>5][j = l&31] = P^T[key pair of r][query]
//                 -- exactly that accumulator register.  The A operand V^T[d][key] is read from the V tile in LDS.
// Every lane owns ONE query column: the online-softmax max/sum/rescale are per-lane scalars (+ one exchange with
// lane^32, which holds the other half of the keys).  One workgroup = 4 waves = 128 queries; K/V tiles of 64 keys in LDS.
//
// Roofline: 4*T^2*d flops per head (fp32 MFMA peak 157.3 TFLOP/s); K/V re-read T/64 times from L2 (1.2 MB per head).
#include "attn_full.h"

namespace ixtts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// K tile pitch (floats): rows stay 16-byte aligned (ds_write_b128 / ds_read_b128) and 8 consecutive keys start 17
// granules apart -> the 8 lanes a b128 read serves per cycle hit 8 different granules mod 8: conflict-free
constexpr int AF_KP = AF_D + 4;

// The contraction index of S^T = K Q^T is free to permute: k-step kk = 4m + i of lane half lh uses d = 8m + 4 lh + i, so a
// lane's four consecutive k-steps are one 16-byte LDS read of K (and one 16-byte global read of Q).
// KSPLIT > 1: blockIdx.z owns a contiguous share of the key tiles and leaves an un-normalised partial; with
// T = 2322 a (batch, head) has only 73 32-query wave tasks -- 1184 waves for 1024 SIMDs -- so one wave per SIMD cannot hide
// its own LDS / softmax / staging latencies (r01 PMC: 0.88 resident waves per SIMD on average, MFMA busy 40 %).
template <int KSPLIT>
__global__ __launch_bounds__(AF_WAVES * 64) void attn_full_f32_kernel(AttnFullArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[AF_KT * AF_KP];
  __shared__ __attribute__((aligned(16))) float Vs[AF_KT * AF_D];
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int q0 = blockIdx.x * (AF_WAVES * AF_QW) + wave * AF_QW;
  const float* qb = a.q + b * a.sb + h * a.sh;
  const float* kb = a.k + b * a.sb + h * a.sh;
  const float* vb = a.v + b * a.sb + h * a.sh;

  // Q^T as the B operand of S^T = K Q^T: lane holds Q[query l31][d = 8m + 4 lh + i] for k-step 4m + i, pre-scaled
  float qr[AF_D / 2];
  {
    const int qi = min(q0 + l31, a.T - 1);
    const float* qp = qb + (long)qi * a.st + 4 * lh;
    // scores are kept in the log2 domain (q pre-scaled by scale * log2 e): softmax weights are one v_exp_f32 each
    const float qs = a.scale * 1.4426950408889634f;
#pragma unroll
    for (int m = 0; m < AF_D / 8; ++m) {
      const float4 t = *reinterpret_cast<const float4*>(qp + 8 * m);
      qr[4 * m] = t.x * qs; qr[4 * m + 1] = t.y * qs; qr[4 * m + 2] = t.z * qs; qr[4 * m + 3] = t.w * qs;
    }
  }
  f32x16 ot[2];  // O^T tiles: d 0..31, 32..63 (rows d in registers, column = this lane's query)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[j][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int n_tiles = (a.T + AF_KT - 1) / AF_KT;
  const int tiles_per = (n_tiles + KSPLIT - 1) / KSPLIT;
  const int t_lo = (KSPLIT > 1 ? blockIdx.z * tiles_per : 0) * AF_KT;
  const int t_hi = KSPLIT > 1 ? min(a.T, t_lo + tiles_per * AF_KT) : a.T;
  for (int t0 = t_lo; t0 < t_hi; t0 += AF_KT) {
    __syncthreads();
    // stage K and V tiles (64 keys x 64 dims): each thread moves 4 float4 of each.  (Measured r01: issuing these loads one
    // tile ahead and holding them in registers across the MFMAs is slower, 442 vs 362 us with 2-wave workgroups -- 64 more
    // live VGPRs.  4-wave workgroups halve the staging per flop; un-split they leave 304 workgroups for 256 CUs (404 us),
    // with the key range split 4 ways they win: 247 us, MFMA busy 64 %.)
#pragma unroll
    for (int i = 0; i < (AF_KT * AF_D / 4) / (AF_WAVES * 64); ++i) {
      const int idx = threadIdx.x + i * (AF_WAVES * 64);
      const int key = idx >> 4, c4 = idx & 15;
      const int t = min(t0 + key, a.T - 1);  // rows beyond T repeat the last one (finite); their scores are masked below
      const float4 kv = *reinterpret_cast<const float4*>(kb + (long)t * a.st + c4 * 4);
      const float4 vv = *reinterpret_cast<const float4*>(vb + (long)t * a.st + c4 * 4);
      *reinterpret_cast<float4*>(Ks + key * AF_KP + c4 * 4) = kv;
      *reinterpret_cast<float4*>(Vs + key * AF_D + c4 * 4) = vv;
    }
    __syncthreads();

    // ---- S^T = K Q^T for the two 32-key halves of the tile; A operands one 16-byte read per four k-steps, read one
    //      group ahead of the MFMAs that consume them
    f32x16 st[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[j][r] = 0.f;
    const float* kp0 = Ks + l31 * AF_KP + 4 * lh;
    const float* kp1 = kp0 + 32 * AF_KP;
    float4 ka[2][2];
    ka[0][0] = *reinterpret_cast<const float4*>(kp0);
    ka[0][1] = *reinterpret_cast<const float4*>(kp1);
#pragma unroll
    for (int m = 0; m < AF_D / 8; ++m) {
      if (m + 1 < AF_D / 8) {
        ka[(m + 1) & 1][0] = *reinterpret_cast<const float4*>(kp0 + 8 * (m + 1));
        ka[(m + 1) & 1][1] = *reinterpret_cast<const float4*>(kp1 + 8 * (m + 1));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 kk4 = ka[m & 1][j];
        st[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk4.x, qr[4 * m], st[j], 0, 0, 0);
        st[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk4.y, qr[4 * m + 1], st[j], 0, 0, 0);
        st[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk4.z, qr[4 * m + 2], st[j], 0, 0, 0);
        st[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk4.w, qr[4 * m + 3], st[j], 0, 0, 0);
      }
    }
    // ---- online softmax for this lane's query; keys beyond T are masked (only the last tile has any)
    if (t0 + AF_KT > a.T) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= a.T) st[j][r] = -INFINITY;
        }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, st[j][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mn = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - mn);  // 0 on the first tile (m = -inf)
    m_run = mn;
    if (alpha != 1.0f) {  // (per lane; a stable running maximum leaves O^T untouched)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[j][r] *= alpha;
    }
    // the weights themselves are computed inside the PV loop below, each right before the MFMA step that consumes it, so
    // that the 32 quarter-rate v_exp_f32 run under the matrix pipe instead of in front of it
    float psum = 0.f;
    // ---- O^T += V^T P^T : k-step (j, r) pairs key (r&3)+8(r>>2) [lanes 0-31] with that key + 4 [lanes 32-63];
    //      the two A operands of a step are read one step ahead
    const float* vp = Vs + (4 * lh) * AF_D + l31;
    float va[2][2];
    va[0][0] = vp[0];
    va[0][1] = vp[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int j = s >> 4, r = s & 15;
      if (s + 1 < 32) {
        const int j1 = (s + 1) >> 4, r1 = (s + 1) & 15;
        const float* vrow = vp + (j1 * 32 + (r1 & 3) + 8 * (r1 >> 2)) * AF_D;
        va[(s + 1) & 1][0] = vrow[0];
        va[(s + 1) & 1][1] = vrow[32];
      }
      const float pv = __builtin_amdgcn_exp2f(st[j][r] - mn);  // B[k = key][j = query]
      psum += pv;
      ot[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s & 1][0], pv, ot[0], 0, 0, 0);
      ot[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s & 1][1], pv, ot[1], 0, 0, 0);
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
  }
  const int qi = q0 + l31;
  if (qi < a.T) {
    if constexpr (KSPLIT == 1) {
      // ---- normalise and store: lane's query column, d rows in registers
      const float inv = 1.0f / l_run;
      float* op = a.o + b * a.osb + (long)qi * a.ost + h * a.osh;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) op[j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = ot[j][r] * inv;
    } else {
      const long row = ((long)blockIdx.z * a.B * a.H + bh) * a.T + qi;
      float* op = a.ws_o + row * AF_D;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) op[j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = ot[j][r];
      if (lh == 0) *reinterpret_cast<float2*>(a.ws_ml + row * 2) = make_float2(m_run, l_run);
    }
  }
}

// out[b,t,h,:] = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M): one wavefront per (b, h, t) row, lane = dim
template <int KSPLIT>
__global__ __launch_bounds__(256) void attn_full_merge_kernel(AttnFullArgs a) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b*H + h)*T + t
  const long rows = (long)a.B * a.H * a.T;
  if (row >= rows) return;
  const int d = threadIdx.x & 63;
  const int t = (int)(row % a.T);
  const int bh = (int)(row / a.T), b = bh / a.H, h = bh % a.H;
  float2 ml[KSPLIT];
  float o[KSPLIT];
  float M = -INFINITY;
#pragma unroll
  for (int sp = 0; sp < KSPLIT; ++sp) {
    ml[sp] = *reinterpret_cast<const float2*>(a.ws_ml + ((long)sp * rows + row) * 2);
    o[sp] = a.ws_o[((long)sp * rows + row) * AF_D + d];
    M = fmaxf(M, ml[sp].x);
  }
  float L = 0.f, O = 0.f;
#pragma unroll
  for (int sp = 0; sp < KSPLIT; ++sp) {
    const float w = (ml[sp].x > -INFINITY) ? __builtin_amdgcn_exp2f(ml[sp].x - M) : 0.f;  // a split without keys has m = -inf, l = 0
    L = fmaf(ml[sp].y, w, L);
    O = fmaf(o[sp], w, O);
  }
  a.o[b * a.osb + (long)t * a.ost + h * a.osh + d] = O / L;
}

}  // namespace ixtts

static size_t attn_partials_bytes(int B, int H, int T) {
  return (size_t)(ixtts::AF_KSPLIT > ixtts::AX_KSPLIT ? ixtts::AF_KSPLIT : ixtts::AX_KSPLIT) * B * H * T * (ixtts::AF_D + 2) * sizeof(float);
}

extern "C" size_t ixtts_attn_full_workspace_bytes(int B, int H, int T) {
  if (B <= 0 || H <= 0 || T <= 0) return 0;
  return attn_partials_bytes(B, H, T) + ixtts::attn_full_x3_plane_bytes(B, H, T);
}

extern "C" int ixtts_attn_full_f32(const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev, int B, int H, int T,
                                   int head_dim, long stride_b, long stride_t, long stride_h, long ostride_b, long ostride_t,
                                   long ostride_h, float scale, void* workspace_dev, size_t workspace_bytes, void* stream) {
  using namespace ixtts;
  IX_ARG(q_dev && k_dev && v_dev && out_dev, "attn_full: null pointer");
  IX_ARG(head_dim == AF_D, "attn_full: head_dim %d (only 64 is built)", head_dim);
  IX_ARG(B > 0 && H > 0 && T > 0 && (long)B * H <= 65535, "attn_full: bad shape B=%d H=%d T=%d", B, H, T);
  IX_ARG(stride_t % 4 == 0 && stride_h % 4 == 0 && stride_b % 4 == 0, "attn_full: strides must keep 16-byte row alignment");
  AttnFullArgs a;
  a.q = q_dev; a.k = k_dev; a.v = v_dev; a.o = out_dev;
  a.sb = stride_b; a.st = stride_t; a.sh = stride_h;
  a.osb = ostride_b; a.ost = ostride_t; a.osh = ostride_h;
  a.B = B; a.H = H; a.T = T; a.scale = scale;
  a.ws_o = a.ws_ml = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int qblocks = ceil_div(T, AF_WAVES * AF_QW);
  // split the keys when the un-split grid cannot give every SIMD two waves and the caller brought the workspace
  const bool have_ws = workspace_dev && workspace_bytes >= ixtts_attn_full_workspace_bytes(B, H, T);
  const bool split = have_ws && T > 4 * AF_KT && (long)qblocks * B * H * AF_WAVES < 2 * 1024;
  // default arithmetic: six bf16 MFMA partial products of exactly split operands (attn_full_x3.hip); IXTTS_ATTN_FULL=f32, or a
  // caller without the workspace for the operand planes: the fp32-MFMA kernel below
  const char* mode = getenv("IXTTS_ATTN_FULL");  // (read per call: the parity tests run both builds in one process)
  const bool x3 = !(mode && strcmp(mode, "f32") == 0);
  const bool use_x3 = x3 && have_ws;
  if (split) {
    a.ws_o = reinterpret_cast<float*>(workspace_dev);
    a.ws_ml = a.ws_o + (size_t)(use_x3 ? AX_KSPLIT : AF_KSPLIT) * B * H * T * AF_D;
  }
  if (use_x3) {
    IX_TRY(launch_attn_full_x3(a, reinterpret_cast<char*>(workspace_dev) + attn_partials_bytes(B, H, T), split, st));
    if (split) {
      const long rows = (long)B * H * T;
      hipLaunchKernelGGL(attn_full_merge_kernel<AX_KSPLIT>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a);
    }
    IX_HIP(hipGetLastError());
    return IXTTS_OK;
  }
  if (split) {
    hipLaunchKernelGGL(attn_full_f32_kernel<AF_KSPLIT>, dim3(qblocks, B * H, AF_KSPLIT), dim3(AF_WAVES * 64), 0, st, a);
    const long rows = (long)B * H * T;
    hipLaunchKernelGGL(attn_full_merge_kernel<AF_KSPLIT>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(attn_full_f32_kernel<1>, dim3(qblocks, B * H), dim3(AF_WAVES * 64), 0, st, a);
  }
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}
