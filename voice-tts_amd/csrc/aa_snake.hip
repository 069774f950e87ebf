// aa_snake.hip -- fused anti-aliased SnakeBeta activation for gfx950 (rows K1 / V4).
//
//   y[t] = sum_k f[k] * s[clamp(2t+k-5, 0, 2T-1)]                         (Down2, pad 5/6)
//   s[j] = u[j] + sin(u[j]*e^alpha)^2 / (e^beta + 1e-9)                    (SnakeBeta, log-scale)
//   u[2m] = 2*sum_a f[11-2a]*x[clamp(m-3+a)] ; u[2m+1] = 2*sum_a f[10-2a]*x[clamp(m-2+a)]   (Up2)
//
// Reference: alias_free_activation/torch/{act.py:24-30,resample.py:29-38,filter.py:92-101},
// activations.py:107-120; the reference's CUDA twin is cuda/anti_alias_activation_cuda.cu:43-179
// (one thread = 32 outputs held in ~180 registers, no LDS).  This is a different design:
// one workgroup = one TILE of one (b,c) row; the x halo tile and the 2x-rate Snake tile
// live in LDS so every sin() is evaluated once, reads/writes of x/y are coalesced 4-byte
// streams, and the down-filter reads LDS as conflict-free 8-byte pairs.
//
// Roofline: 8 B of HBM per element (read x, write y); 2 sin^2 (12 instructions each) + 24 FMA per element.
#include "common.h"

namespace ixtts {

constexpr int AA_TILE = 1024;    // outputs per workgroup
constexpr int AA_THREADS = 256;  // 4 waves: four consecutive outputs per thread
constexpr int AA_XH = 10;        // x halo each side (7 needed; 10 puts a thread's first input on a 16-byte LDS boundary)
constexpr int AA_NX = AA_TILE + 2 * AA_XH;
constexpr int AA_NS = 2 * AA_TILE + 16;  // ss[i] = s[2*t0 - 6 + i]
static_assert(AA_TILE == 4 * AA_THREADS, "four outputs per thread");

// sin(z)^2 for the Snake term.  sin^2 has period pi and is even, so one Cody-Waite reduction to r = z - n*pi in
// [-pi/2, pi/2] (two fused steps, pi split hi + lo) and an odd minimax polynomial for sin(r) (degree 11) do it with no
// quadrant logic: 12 instructions against ~40 for sinf(), 3e-7 absolute on sin^2 for |z| up to a few thousand -- the rounding
// floor of squaring a float sine.  (The two sines were 70 % of this kernel's instructions.)
__device__ __forceinline__ float sin_squared(float z) {
  const float n = rintf(z * 0.318309886183790672f);
  float r = fmaf(-n, 3.14159274101257324f, z);
  r = fmaf(n, 8.74227765734758577e-08f, r);  // pi - float(pi) = -8.74e-8
  const float w = r * r;
  float p = -2.377017516153046e-08f;
  p = fmaf(p, w, 2.7517328362591797e-06f);
  p = fmaf(p, w, -0.00019840669119730592f);
  p = fmaf(p, w, 0.008333329111337662f);
  p = fmaf(p, w, -0.1666666716337204f);
  p = fmaf(p, w, 1.0f);
  const float sn = r * p;
  return sn * sn;
}

template <bool FAST_SIN>
__device__ __forceinline__ float snake(float u, float a, float inv_b) {
  if constexpr (FAST_SIN) {
    const float sn = __sinf(u * a);
    return u + inv_b * sn * sn;
  } else {
    return u + inv_b * sin_squared(u * a);
  }
}

template <bool FAST_SIN>
__global__ __launch_bounds__(AA_THREADS) void aa_snake_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               const float* __restrict__ up12,
                                                               const float* __restrict__ down12,
                                                               const float* __restrict__ log_alpha,
                                                               const float* __restrict__ log_beta, int C, int T) {
  __shared__ __attribute__((aligned(16))) float xs[AA_NX];
  __shared__ __attribute__((aligned(16))) float ss[AA_NS];

  const int row = blockIdx.y;  // b*C + c
  const int c = row % C;
  const int t0 = blockIdx.x * AA_TILE;
  const float* xr = x + (size_t)row * T;
  float* yr = y + (size_t)row * T;

  float fu[12], fd[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    fu[k] = up12[k];
    fd[k] = down12[k];
  }
  const float a = expf(log_alpha[c]);
  const float inv_b = 1.0f / (expf(log_beta[c]) + 1e-9f);

  // stage x[t0-10 .. t0+TILE+10) with replicate clamping
  for (int i = threadIdx.x; i < AA_NX; i += AA_THREADS) {
    int t = t0 - AA_XH + i;
    t = min(max(t, 0), T - 1);
    xs[i] = xr[t];
  }
  __syncthreads();

  // pair p <-> m = t0-3+p owns s[2m] -> ss[2p] and s[2m+1] -> ss[2p+1]; both share the 7 staged inputs x[m-3..m+3]
  auto pair = [&](const float* xw, float& se, float& so) {
    float ue = 0.f, uo = 0.f;
#pragma unroll
    for (int aa = 0; aa < 6; ++aa) {
      ue = fmaf(fu[11 - 2 * aa], xw[aa], ue);      // u[2m]   = 2 * sum_a f[11-2a] x[m-3+a]
      uo = fmaf(fu[10 - 2 * aa], xw[aa + 1], uo);  // u[2m+1] = 2 * sum_a f[10-2a] x[m-2+a]
    }
    se = snake<FAST_SIN>(2.0f * ue, a, inv_b);
    so = snake<FAST_SIN>(2.0f * uo, a, inv_b);
  };
  const bool interior = t0 >= 3 && t0 + AA_TILE + 3 <= T - 1;  // every pair's m inside [0, T-1]: no replicate padding of s
  if (interior) {
    // phase 1: four consecutive pairs per thread from ten staged inputs (two 16-byte + one 8-byte LDS read, two 16-byte writes)
    const int p0 = 4 * threadIdx.x;
    float X[10];
    *reinterpret_cast<float4*>(X) = *reinterpret_cast<const float4*>(xs + p0 + 4);  // x[m-3] of pair p0 sits at xs[p0 + 4]
    *reinterpret_cast<float4*>(X + 4) = *reinterpret_cast<const float4*>(xs + p0 + 8);
    *reinterpret_cast<float2*>(X + 8) = *reinterpret_cast<const float2*>(xs + p0 + 12);
    float S[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) pair(X + q, S[2 * q], S[2 * q + 1]);
    *reinterpret_cast<float4*>(ss + 2 * p0) = *reinterpret_cast<const float4*>(S);
    *reinterpret_cast<float4*>(ss + 2 * p0 + 4) = *reinterpret_cast<const float4*>(S + 4);
    if (threadIdx.x < 7) {  // the seven pairs past the tile
      const int pr = AA_TILE + threadIdx.x;
      float xw[7], se, so;
#pragma unroll
      for (int i = 0; i < 7; ++i) xw[i] = xs[pr + 4 + i];
      pair(xw, se, so);
      ss[2 * pr] = se;
      ss[2 * pr + 1] = so;
    }
  } else {
    // first / last tile of a row: replicate padding of s (m < 0 -> s[0], m > T-1 -> s[2T-1]), one pair per thread-iteration
    for (int pr = threadIdx.x; pr < AA_TILE + 7; pr += AA_THREADS) {
      const int m = t0 - 3 + pr;
      const int mc = min(max(m, 0), T - 1);
      const float* xp = xs + (mc - 3) - (t0 - AA_XH);  // x[mc-3] .. x[mc+3] (xs is already replicate-clamped)
      float xw[7], se, so;
#pragma unroll
      for (int i = 0; i < 7; ++i) xw[i] = xp[i];
      pair(xw, se, so);
      if (m < 0) so = se;
      if (m > T - 1) se = so;
      ss[2 * pr] = se;
      ss[2 * pr + 1] = so;
    }
  }
  __syncthreads();

  // phase 2: y[t0+o] = sum_k fd[k] * s[2(t0+o)-5+k] = sum_k fd[k] * ss[2o+1+k]; four consecutive outputs per thread from
  // twenty values (five 16-byte LDS reads)
  {
    const int o0 = 4 * threadIdx.x;
    float S[20];
#pragma unroll
    for (int i = 0; i < 5; ++i) *reinterpret_cast<float4*>(S + 4 * i) = *reinterpret_cast<const float4*>(ss + 2 * o0 + 4 * i);
    float acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q] = 0.f;
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[q] = fmaf(fd[k], S[2 * q + 1 + k], acc[q]);
    }
    const int t = t0 + o0;
    if (t + 3 < T && ((reinterpret_cast<uintptr_t>(yr + t) & 15) == 0)) {
      *reinterpret_cast<float4*>(yr + t) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (t + q < T) yr[t + q] = acc[q];
    }
  }
}

// ---- the same activation writing the conv kernel's x planes (conv1d_x3.hip): y as three bf16 pieces per value, eight
// consecutive channels of a time step in one 16-byte unit -> Xp[b][plane][C/8][T].  One workgroup = one channel octet x 256
// outputs; wave r runs channel 8*o + r exactly as the fp32 kernel does (four consecutive outputs per lane), the 8 x 256 results
// are transposed through LDS, and thread t splits and packs the eight channels of time step t: three 16-byte stores.
constexpr int AP_TILE = 256;
constexpr int AP_NX = AP_TILE + 2 * AA_XH;
constexpr int AP_NS = 2 * AP_TILE + 16;


#ifndef IXTTS_AP_NSUB
#define IXTTS_AP_NSUB 4
#endif
constexpr int AP_NSUB = IXTTS_AP_NSUB;  // consecutive 256-output tiles per workgroup: tile s+1's inputs are fetched under tile s's arithmetic

template <bool FAST_SIN>
__global__ __launch_bounds__(512) void aa_snake_planes_kernel(const float* __restrict__ x, uint4* __restrict__ xp, const float* __restrict__ up12,
                                                              const float* __restrict__ down12, const float* __restrict__ log_alpha,
                                                              const float* __restrict__ log_beta, int C, int T) {
  __shared__ __attribute__((aligned(16))) float xs[8][AP_NX];
  __shared__ __attribute__((aligned(16))) float ss[8][AP_NS];
  __shared__ __attribute__((aligned(16))) float ys[2][8][AP_TILE];  // (two: tile s+1's results are written while tile s's are packed)

  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;  // wave r = channel r of the octet
  const int o = blockIdx.y, b = blockIdx.z;
  const int C8 = (C + 7) >> 3;
  const int c = min(o * 8 + r, C - 1);  // (an octet past the tensor's channels: its rows are written as zeros below)
  const bool live = o * 8 + r < C;
  const float* xr = x + ((size_t)b * C + c) * T;
  constexpr int NXR = (AP_NX + 63) / 64;  // staged inputs per lane and tile

  // a short-lived workgroup (one tile) spent its life waiting for its own first loads with only 2-4 workgroups per CU to hide it:
  // a workgroup now walks AP_NSUB tiles and holds the next tile's inputs in registers while it works on the current one
  float xn[NXR];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
      const int t = min(max(t0 - AA_XH + lane + 64 * j, 0), T - 1);
      xn[j] = xr[t];
    }
  };
  const int tile0 = blockIdx.x * AP_NSUB;
  fetch(tile0 * AP_TILE);

  float fu[12], fd[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    fu[k] = up12[k];
    fd[k] = down12[k];
  }
  const float a = expf(log_alpha[c]);
  const float inv_b = 1.0f / (expf(log_beta[c]) + 1e-9f);
  auto pair = [&](const float* xw, float& se, float& so) {
    float ue = 0.f, uo = 0.f;
#pragma unroll
    for (int aa = 0; aa < 6; ++aa) {
      ue = fmaf(fu[11 - 2 * aa], xw[aa], ue);
      uo = fmaf(fu[10 - 2 * aa], xw[aa + 1], uo);
    }
    se = snake<FAST_SIN>(2.0f * ue, a, inv_b);
    so = snake<FAST_SIN>(2.0f * uo, a, inv_b);
  };

  for (int sub = 0; sub < AP_NSUB; ++sub) {
    const int t0 = (tile0 + sub) * AP_TILE;
    if (t0 >= T) break;  // (uniform)
#pragma unroll
    for (int j = 0; j < NXR; ++j)
      if (lane + 64 * j < AP_NX) xs[r][lane + 64 * j] = xn[j];
    if (sub + 1 < AP_NSUB && t0 + AP_TILE < T) fetch(t0 + AP_TILE);
    // xs[r] and ss[r] are private to wave r (a wave's LDS accesses execute in order): no workgroup barrier between the staging, the
    // up-filter + Snake and the down-filter -- the eight waves drift apart and one's loads sit under another's arithmetic.  The one
    // exchange between waves is the 8-channel transpose for the packed planes below.
    const bool interior = t0 >= 3 && t0 + AP_TILE + 3 <= T - 1;
    if (interior) {
      const int p0 = 4 * lane;
      float X[10];
      *reinterpret_cast<float4*>(X) = *reinterpret_cast<const float4*>(&xs[r][p0 + 4]);
      *reinterpret_cast<float4*>(X + 4) = *reinterpret_cast<const float4*>(&xs[r][p0 + 8]);
      *reinterpret_cast<float2*>(X + 8) = *reinterpret_cast<const float2*>(&xs[r][p0 + 12]);
      float S[8];
#ifdef IXTTS_SNK_NOMATH
#pragma unroll
      for (int q = 0; q < 4; ++q) S[2 * q] = X[q + 3], S[2 * q + 1] = X[q + 3];
#else
#pragma unroll
      for (int q = 0; q < 4; ++q) pair(X + q, S[2 * q], S[2 * q + 1]);
#endif
      *reinterpret_cast<float4*>(&ss[r][2 * p0]) = *reinterpret_cast<const float4*>(S);
      *reinterpret_cast<float4*>(&ss[r][2 * p0 + 4]) = *reinterpret_cast<const float4*>(S + 4);
      if (lane < 7) {
        const int pr = AP_TILE + lane;
        float xw[7], se, so;
#pragma unroll
        for (int i = 0; i < 7; ++i) xw[i] = xs[r][pr + 4 + i];
        pair(xw, se, so);
        ss[r][2 * pr] = se;
        ss[r][2 * pr + 1] = so;
      }
    } else {
      for (int pr = lane; pr < AP_TILE + 7; pr += 64) {
        const int m = t0 - 3 + pr;
        const int mc = min(max(m, 0), T - 1);
        const float* xq = &xs[r][(mc - 3) - (t0 - AA_XH)];
        float xw[7], se, so;
#pragma unroll
        for (int i = 0; i < 7; ++i) xw[i] = xq[i];
        pair(xw, se, so);
        if (m < 0) so = se;
        if (m > T - 1) se = so;
        ss[r][2 * pr] = se;
        ss[r][2 * pr + 1] = so;
      }
    }
    {
      const int o0 = 4 * lane;
      float S[20];
#pragma unroll
      for (int i = 0; i < 5; ++i) *reinterpret_cast<float4*>(S + 4 * i) = *reinterpret_cast<const float4*>(&ss[r][2 * o0 + 4 * i]);
      float acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q] = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[q] = fmaf(fd[k], S[2 * q + 1 + k], acc[q]);
        if (!live) acc[q] = 0.f;
      }
      *reinterpret_cast<float4*>(&ys[sub & 1][r][o0]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();  // the one barrier of a tile (ys[sub & 1] is rewritten two tiles later: behind the next tile's barrier)
    if (threadIdx.x < AP_TILE && t0 + (int)threadIdx.x < T) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ys[sub & 1][e][threadIdx.x];
      uint4 ph, pm, pl;
      split8_bf16x3(v, ph, pm, pl);
      uint4* dst = xp + ((size_t)b * 3 * C8 + o) * T + t0 + threadIdx.x;
#ifdef IXTTS_SNK_NOSTORE
      if (ph.x == 0x12345678u && T < 0) {
#endif
      dst[0] = ph;
      dst[(size_t)C8 * T] = pm;
      dst[(size_t)2 * C8 * T] = pl;
#ifdef IXTTS_SNK_NOSTORE
      }
#endif
    }
  }
}

int launch_aa_snake_planes(const float* x, void* xplanes, const float* up12, const float* down12, const float* la, const float* lb, int B, int C, int T,
                           bool fast_sin, hipStream_t st) {
  if (B * C == 0 || T == 0) return IXTTS_OK;
  IX_ARG(B > 0 && C > 0 && T > 0, "aa_snake_planes: bad shape B=%d C=%d T=%d", B, C, T);
  IX_ARG((C + 7) / 8 <= 65535 && B <= 65535, "aa_snake_planes: grid too large");
  dim3 grid(ceil_div(T, AP_TILE * AP_NSUB), (C + 7) / 8, B);
  if (fast_sin)
    hipLaunchKernelGGL(aa_snake_planes_kernel<true>, grid, dim3(512), 0, st, x, reinterpret_cast<uint4*>(xplanes), up12, down12, la, lb, C, T);
  else
    hipLaunchKernelGGL(aa_snake_planes_kernel<false>, grid, dim3(512), 0, st, x, reinterpret_cast<uint4*>(xplanes), up12, down12, la, lb, C, T);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

int launch_aa_snake(const float* x, float* y, const float* up12, const float* down12, const float* la,
                    const float* lb, int B, int C, int T, bool fast_sin, hipStream_t st) {
  if (B * C == 0 || T == 0) return IXTTS_OK;
  IX_ARG(B > 0 && C > 0 && T > 0, "aa_snake: bad shape B=%d C=%d T=%d", B, C, T);
  IX_ARG((long long)B * C <= 65535, "aa_snake: B*C=%lld exceeds grid.y", (long long)B * C);
  dim3 grid(ceil_div(T, AA_TILE), B * C);
  if (fast_sin)
    hipLaunchKernelGGL(aa_snake_kernel<true>, grid, dim3(AA_THREADS), 0, st, x, y, up12, down12, la, lb, C, T);
  else
    hipLaunchKernelGGL(aa_snake_kernel<false>, grid, dim3(AA_THREADS), 0, st, x, y, up12, down12, la, lb, C, T);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

}  // namespace ixtts

extern "C" int ixtts_aa_snake_f32(const float* x_dev, float* y_dev, const float* up12_dev, const float* down12_dev,
                                  const float* log_alpha_dev, const float* log_beta_dev, int B, int C, int T,
                                  void* stream) {
  using namespace ixtts;
  IX_ARG(B >= 0 && C >= 0 && T >= 0, "aa_snake: negative shape");
  if (B == 0 || C == 0 || T == 0) return IXTTS_OK;
  IX_ARG(x_dev && y_dev && up12_dev && down12_dev && log_alpha_dev && log_beta_dev, "aa_snake: null pointer");
  IX_ARG(x_dev != y_dev, "aa_snake: in-place call not supported");
  return launch_aa_snake(x_dev, y_dev, up12_dev, down12_dev, log_alpha_dev, log_beta_dev, B, C, T, false,
                         (hipStream_t)stream);
}
