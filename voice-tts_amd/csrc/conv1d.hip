// conv1d.hip -- BigVGAN Conv1d / ConvTranspose1d as an implicit GEMM on the fp32 matrix
// cores of gfx950 (rows V1, V2, V3 of SURVEY.md 8(a)).
//
//   y[b, co, q*os + oo] = bias[co] + sum_{tap} sum_{ci} Wp[tap][ci][co] * x[b, ci, q + off0 + tap*dil]
//                         (+ res[b,co,t]) (+ accum[b,co,t]) (/ div)
//
// A dilated "same" Conv1d is os=1, oo=0, off0=-pad.  A ConvTranspose1d of stride s is run
// as s polyphase convolutions (phase r: taps k = r + s*j, input index q - j, output
// t = q*s + r - pad), selected by blockIdx.z.
//
// Reference: torch Conv1d/ConvTranspose1d call sites bigvgan.py:56-88,132-141,285-316,348-350
// (cuDNN on the reference's GPU path).  Design here is MI355X-first:
//   * MFMA `v_mfma_f32_32x32x2_f32`: exact fp32 (a k-ordered fmaf chain), 157 TFLOP/s peak --
//     the vocoder must stay fp32 for the 1e-3 waveform budget and the fused network is
//     compute-bound, so this is the binding roofline.
//   * weights are pre-packed [tap][ci][co] so a 32-lane half-wave reads 128 contiguous
//     bytes per A fragment; the x tile (+dilation halo) is staged once per K-chunk in LDS
//     and every tap re-reads it at a shifted offset (no im2col in memory).
//   * bias / residual add / 3-way resblock accumulate (/3) are fused into the epilogue.
//   * blockIdx -> tile mapping is XCD-aware: the 8 XCDs each take a contiguous run of
//     tiles (same co rows), so one XCD's L2 holds one slice of the weights.
#include "conv.h"

namespace ixtts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CK = 8;  // input channels per K-chunk

template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(ConvParams p) {
  constexpr int BM = 32 * MT * WM;
  constexpr int BN = 32 * NT * WN;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // ---- XCD-aware tile id (blocks b and b+8 share an XCD; give each XCD a contiguous run)
  const int nwg = p.n_tiles * p.m_tiles;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int lin = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
  const int m_tile = lin / p.n_tiles;
  const int n_tile = lin - m_tile * p.n_tiles;

  const int phase = blockIdx.z;
  const int b = blockIdx.y;
  const int m0 = m_tile * BM;
  const int q0 = n_tile * BN;
  const int adil = p.dil < 0 ? -p.dil : p.dil;
  const int span = (p.ntap - 1) * adil;
  const int XW = BN + span;
  const int XWP = XW | 1;  // odd row pitch
  float* Xs = smem;                 // [CK][XWP]
  float* Ws = smem + CK * XWP;      // [ntap][CK][BM]
  Ws = (float*)(((uintptr_t)Ws + 15) & ~(uintptr_t)15);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* xb = p.x + (size_t)b * p.Cin * p.Tin;
  const float* wph = p.wp + (size_t)phase * p.ntap * p.Cin_pad * p.Cout_pad;
  // smallest input index any tap of this tile touches
  const int lo = q0 + p.off0 - (p.dil < 0 ? span : 0);

  for (int c0 = 0; c0 < p.Cin_pad; c0 += CK) {
    __syncthreads();
    // ---- stage x tile: CK rows x XW columns (zero outside [0,Tin) / beyond Cin)
#pragma unroll
    for (int ci = 0; ci < CK; ++ci) {
      const bool crow = (c0 + ci) < p.Cin;
      const float* xr = xb + (size_t)(c0 + ci) * p.Tin;
      for (int j = tid; j < XW; j += 256) {
        int t = lo + j;
        float v = 0.f;
        if (crow && t >= 0 && t < p.Tin) v = xr[t];
        Xs[ci * XWP + j] = v;
      }
    }
    // ---- stage weight tile: ntap*CK rows of BM floats
    {
      constexpr int V4 = BM / 4;
      const int rows = p.ntap * CK;
      for (int i = tid; i < rows * V4; i += 256) {
        int row = i / V4, c4 = i - row * V4;
        int tap = row / CK, ci = row - tap * CK;
        const float4 v = *reinterpret_cast<const float4*>(wph + ((size_t)tap * p.Cin_pad + c0 + ci) * p.Cout_pad + m0 + c4 * 4);
        *reinterpret_cast<float4*>(Ws + row * BM + c4 * 4) = v;
      }
    }
    __syncthreads();
    // ---- MFMA over taps x channel pairs
    for (int tap = 0; tap < p.ntap; ++tap) {
      const int xoff = (p.dil >= 0) ? tap * adil : (p.ntap - 1 - tap) * adil;
      const float* wrow = Ws + (tap * CK + lh) * BM + wm * (32 * MT) + l31;
      const float* xrow = Xs + lh * XWP + wn * (32 * NT) + l31 + xoff;
#pragma unroll
      for (int kk = 0; kk < CK / 2; ++kk) {
        float a[MT], bb[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = wrow[kk * 2 * BM + i * 32];
#pragma unroll
        for (int j = 0; j < NT; ++j) bb[j] = xrow[kk * 2 * XWP + j * 32];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: bias, residual, 3-way accumulate, /3
  const size_t ob = (size_t)b * p.Cout * p.Tout;
  const int ophase = p.oo + (p.nphase > 1 ? phase : 0);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int q = q0 + wn * (32 * NT) + j * 32 + l31;
      const int t = q * p.os + ophase;
      const bool tv = (q < p.Nq) && (t >= 0) && (t < p.Tout);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (tv && m < p.Cout) {
          const size_t o = ob + (size_t)m * p.Tout + t;
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[m];
          if (p.res) v += p.res[o];
          if (p.accum) v = p.accum[o] + v;
          if (p.div3) v = v / 3.0f;
          p.y[o] = v;
        }
      }
    }
  }
}

template <int MT, int NT, int WM, int WN>
static int launch_cfg(const ConvParams& p0, hipStream_t st) {
  ConvParams p = p0;
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  p.m_tiles = ceil_div(p.Cout, BM);
  p.n_tiles = ceil_div(p.Nq, BN);
  IX_ARG(p.Cout_pad % BM == 0 && p.Cout_pad >= p.m_tiles * BM, "conv: Cout_pad %d not a multiple of BM %d", p.Cout_pad, BM);
  int adil = p.dil < 0 ? -p.dil : p.dil;
  int XWP = (BN + (p.ntap - 1) * adil) | 1;
  size_t smem = (size_t)(CK * XWP + 4 + p.ntap * CK * BM) * sizeof(float);
  IX_ARG(smem <= 160 * 1024, "conv: LDS tile %zu B too large", smem);
  auto kern = conv1d_mfma_kernel<MT, NT, WM, WN>;
  if (smem > 64 * 1024) {
    static bool done = false;
    if (!done) {
      IX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done = true;
    }
  }
  dim3 grid(p.m_tiles * p.n_tiles, p.B, p.nphase);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

// tile shape id chosen at pack time from Cout (see conv_tile_bm)
int conv_tile_bm(int Cout) {
  if (Cout <= 32) return 32;
  if (Cout <= 64) return 64;
  if (Cout % 128 != 0 && Cout % 96 == 0) return 96;
  return 128;
}

int launch_conv1d(const ConvParams& p, hipStream_t st) {
  IX_ARG(p.Cin_pad % CK == 0, "conv: Cin_pad %d not a multiple of %d", p.Cin_pad, CK);
  switch (conv_tile_bm(p.Cout)) {
    case 32: return launch_cfg<1, 2, 1, 4>(p, st);   // 32 x 256
    case 64: return launch_cfg<2, 1, 1, 4>(p, st);   // 64 x 128
    case 96: return launch_cfg<3, 1, 1, 4>(p, st);   // 96 x 128
    default: return launch_cfg<2, 2, 2, 2>(p, st);   // 128 x 128
  }
}

// ------------------------------------------------------------------------------------
// conv_post (Cout = 1, k = 7, no bias) + clamp(-1, 1)   (bigvgan.py:348-350,378-384)
// One output per thread; 7*C taps from L1/L2-resident rows.  0.09 GFLOP / 1000 frames.
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int C,
                                                         int T) {
  extern __shared__ float ws[];  // [C][7]
  for (int i = threadIdx.x; i < C * 7; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float* xb = x + (size_t)b * C * T;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* xr = xb + (size_t)c * T;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      int ti = t + k - 3;
      float v = (ti >= 0 && ti < T) ? xr[ti] : 0.f;
      acc = fmaf(ws[c * 7 + k], v, acc);
    }
  }
  if (bias) acc += bias[0];
  y[(size_t)b * T + t] = fminf(fmaxf(acc, -1.0f), 1.0f);
}

int launch_conv_post(const float* x, const float* w, const float* bias, float* y, int B, int C, int T, hipStream_t st) {
  if (B == 0 || T == 0) return IXTTS_OK;
  dim3 grid(ceil_div(T, 256), B);
  hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), C * 7 * sizeof(float), st, x, w, bias, y, C, T);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

}  // namespace ixtts
