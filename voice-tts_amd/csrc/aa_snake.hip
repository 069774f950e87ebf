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
// Roofline: 8 B of HBM per element (read x, write y); 2 sin + ~40 FMA per element.
#include "common.h"

namespace ixtts {

constexpr int AA_TILE = 1024;    // outputs per workgroup
constexpr int AA_THREADS = 256;  // 4 waves
constexpr int AA_XH = 7;         // x halo each side
constexpr int AA_NX = AA_TILE + 2 * AA_XH;
constexpr int AA_NS = 2 * AA_TILE + 12;  // s[2*t0-5 .. 2*(t0+TILE)+6]

template <bool FAST_SIN>
__device__ __forceinline__ float snake(float u, float a, float inv_b) {
  float sn;
  if constexpr (FAST_SIN) {
    sn = __sinf(u * a);
  } else {
    sn = sinf(u * a);
  }
  return u + inv_b * sn * sn;
}

// Shared device routine: also used by the conv kernels' fused prologue.
template <bool FAST_SIN>
__global__ __launch_bounds__(AA_THREADS) void aa_snake_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               const float* __restrict__ up12,
                                                               const float* __restrict__ down12,
                                                               const float* __restrict__ log_alpha,
                                                               const float* __restrict__ log_beta, int C, int T) {
  __shared__ float xs[AA_NX];
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

  // stage x[t0-6 .. t0+TILE+6) with replicate clamping
  for (int i = threadIdx.x; i < AA_NX; i += AA_THREADS) {
    int t = t0 - AA_XH + i;
    t = min(max(t, 0), T - 1);
    xs[i] = xr[t];
  }
  __syncthreads();

  // phase 1: one (even, odd) pair of up-sampled Snake values per thread-iteration.  Pair p <-> m = t0-3+p
  // owns s[2m] -> ss[2p-1] and s[2m+1] -> ss[2p]; both share the 7 staged inputs x[m-3..m+3] (xs is already
  // replicate-clamped, so no per-tap index clamps).  Replicate padding of s: m < 0 -> s[0], m > T-1 -> s[2T-1].
  for (int pr = threadIdx.x; pr < AA_TILE + 7; pr += AA_THREADS) {
    const int m = t0 - 3 + pr;
    const int mc = min(max(m, 0), T - 1);
    const float* xp = xs + (mc - 3) - (t0 - AA_XH);  // x[mc-3] .. x[mc+3]
    float xw[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) xw[i] = xp[i];
    float ue = 0.f, uo = 0.f;
#pragma unroll
    for (int aa = 0; aa < 6; ++aa) {
      ue = fmaf(fu[11 - 2 * aa], xw[aa], ue);      // u[2m]   = 2 * sum_a f[11-2a] x[m-3+a]
      uo = fmaf(fu[10 - 2 * aa], xw[aa + 1], uo);  // u[2m+1] = 2 * sum_a f[10-2a] x[m-2+a]
    }
    float se = snake<FAST_SIN>(2.0f * ue, a, inv_b);
    float so = snake<FAST_SIN>(2.0f * uo, a, inv_b);
    if (m < 0) so = se;
    if (m > T - 1) se = so;
    const int ie = 2 * pr - 1;
    if (ie >= 0 && ie < AA_NS) ss[ie] = se;
    if (ie + 1 < AA_NS) ss[ie + 1] = so;
  }
  __syncthreads();

  // phase 2: y[t0+o] = sum_k fd[k]*ss[2*o+k]  (ss[i] <-> j = 2*t0-5+i)
  for (int o = threadIdx.x; o < AA_TILE; o += AA_THREADS) {
    int t = t0 + o;
    if (t >= T) break;
    const float2* sp = reinterpret_cast<const float2*>(ss + 2 * o);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      float2 v = sp[k];
      acc = fmaf(fd[2 * k], v.x, acc);
      acc = fmaf(fd[2 * k + 1], v.y, acc);
    }
    yr[t] = acc;
  }
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
