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
//   * weights are pre-packed [tap][ci/8][ci%2][co][(ci%8)/2]: one float4 per lane = the A operand of 4
//     k-steps, streamed from L2 into registers one tap ahead (no LDS round trip for weights);
//     the x tile of 32 input channels (+dilation halo) is staged once per K-chunk in LDS and
//     every tap re-reads it at a shifted offset (no im2col in memory).
//   * bias / residual add / 3-way resblock accumulate (/3) are fused into the epilogue.
//   * blockIdx -> tile mapping is XCD-aware: the 8 XCDs each take a contiguous run of
//     tiles (same co rows), so one XCD's L2 holds one slice of the weights.
#include "conv.h"

namespace ixtts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CG = 8;   // input channels per weight group (one float4 per lane = 4 MFMA k-steps)
constexpr int CKG = 4;  // groups per K-chunk: the x tile in LDS covers 32 input channels

// Weights are packed Wq[phase][tap][ci/8][ci%2][co][(ci%8)/2]: lane (co = l31, k-parity = lh) reads ONE float4
// holding its A operand for the 4 consecutive k-steps of a channel group; a half-wave reads 512 contiguous bytes.
// They stream from L2 (the XCD-aware tile map keeps one co-slice per XCD) straight into registers, one tap ahead;
// only the x tile (+dilation halo) lives in LDS.
template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(ConvParams p) {
  constexpr int BM = 32 * MT * WM;
  constexpr int BN = 32 * NT * WN;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) float Xs[];  // [CKG*CG][XWP]

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
  const int ngroups = p.Cin_pad / CG;
  // A operand base of this lane: [.. tap][group][lh][co][4]
  const float4* wq = reinterpret_cast<const float4*>(p.wp) + (size_t)phase * p.ntap * ngroups * 2 * p.Cout_pad +
                     (size_t)lh * p.Cout_pad + m0 + wm * (32 * MT) + l31;
  const size_t gstride = (size_t)2 * p.Cout_pad;       // float4 per group
  const size_t tstride = (size_t)ngroups * gstride;    // float4 per tap
  const int lo = q0 + p.off0 - (p.dil < 0 ? span : 0);  // smallest input index any tap of this tile touches

  for (int g0 = 0; g0 < ngroups; g0 += CKG) {
    const int ng = min(CKG, ngroups - g0);
    __syncthreads();
    // ---- stage x tile: ng*8 rows x XW columns (zero outside [0,Tin) / beyond Cin).  One channel group at a
    // time: its 8 rows x up-to-2 columns per thread are all in flight before the first LDS store (16 temporaries;
    // unrolling the whole 32-row chunk -- or a register-prefetched double buffer -- pushed the kernel past 200 VGPRs
    // and to occupancy 1, which measured slower).
    for (int gg = 0; gg < ng; ++gg) {
      float v[CG][2];
#pragma unroll
      for (int r = 0; r < CG; ++r) {
        const int c = (g0 + gg) * CG + r;
        const float* xr = xb + (size_t)c * p.Tin;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = tid + jj * 256, t = lo + j;
          v[r][jj] = (j < XW && c < p.Cin && t >= 0 && t < p.Tin) ? xr[t] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < CG; ++r)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = tid + jj * 256;
          if (j < XW) Xs[(gg * CG + r) * XWP + j] = v[r][jj];
        }
    }
    __syncthreads();
    const float* Xc = Xs;
    // ---- taps: A fragments one tap ahead in registers, B fragments from the LDS tile at a shifted offset
    float4 a_cur[CKG][MT], a_nxt[CKG][MT];
#pragma unroll
    for (int gg = 0; gg < CKG; ++gg)
#pragma unroll
      for (int i = 0; i < MT; ++i) a_cur[gg][i] = (gg < ng) ? wq[(size_t)(g0 + gg) * gstride + i * 32] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tap = 0; tap < p.ntap; ++tap) {
      if (tap + 1 < p.ntap) {
#pragma unroll
        for (int gg = 0; gg < CKG; ++gg)
#pragma unroll
          for (int i = 0; i < MT; ++i)
            a_nxt[gg][i] = (gg < ng) ? wq[(size_t)(tap + 1) * tstride + (size_t)(g0 + gg) * gstride + i * 32] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const int xoff = (p.dil >= 0) ? tap * adil : (p.ntap - 1 - tap) * adil;
      const float* xrow = Xc + lh * XWP + wn * (32 * NT) + l31 + xoff;
#pragma unroll
      for (int gg = 0; gg < CKG; ++gg) {
        if (gg < ng) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            float bb[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bb[j] = xrow[(gg * CG + 2 * kk) * XWP + j * 32];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const float av = kk == 0 ? a_cur[gg][i].x : kk == 1 ? a_cur[gg][i].y : kk == 2 ? a_cur[gg][i].z : a_cur[gg][i].w;
#pragma unroll
              for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bb[j], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int gg = 0; gg < CKG; ++gg)
#pragma unroll
        for (int i = 0; i < MT; ++i) a_cur[gg][i] = a_nxt[gg][i];
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
  size_t smem = (size_t)(CKG * CG * XWP) * sizeof(float);
  IX_ARG(smem <= 160 * 1024, "conv: LDS tile %zu B too large", smem);
  IX_ARG(BN + (p.ntap - 1) * adil <= 512, "conv: x tile of %d columns exceeds the staging bound (512)", BN + (p.ntap - 1) * adil);
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

// Cout padding granule chosen at pack time (a multiple of every BM that may be used for this Cout)
int conv_tile_bm(int Cout) {
  if (Cout <= 32) return 32;
  if (Cout <= 64) return 64;
  if (Cout % 128 != 0 && Cout % 96 == 0) return 96;
  return 128;
}

// Workgroups of one launch run as ceil(wgs / 256) "rounds" over the 256 CUs (co-resident workgroups of a CU share
// its matrix pipe), so a grid of 375 tiles costs as much as 512.  Score = tile-quantisation efficiency x a mild
// preference for bigger tiles (fewer LDS/L2 bytes per flop).
static double tile_score(const ConvParams& p, int BM, int BN, double pref) {
  const double wgs = (double)ceil_div(p.Cout, BM) * ceil_div(p.Nq, BN) * p.B * p.nphase;
  const double rounds = (double)(((long long)wgs + 255) / 256);
  return wgs / (rounds * 256.0) * pref;
}

int launch_conv1d(const ConvParams& p, hipStream_t st) {
  IX_ARG(p.Cin_pad % CG == 0, "conv: Cin_pad %d not a multiple of %d", p.Cin_pad, CG);
  switch (conv_tile_bm(p.Cout)) {
    case 32: return launch_cfg<1, 2, 1, 4>(p, st);   // 32 x 256
    case 64: return launch_cfg<2, 1, 1, 4>(p, st);   // 64 x 128
    case 96: return launch_cfg<3, 1, 1, 4>(p, st);   // 96 x 128
    default: {
      const double s128 = tile_score(p, 128, 128, 1.00), s64x128 = tile_score(p, 64, 128, 0.96), s64 = tile_score(p, 64, 64, 0.90);
      if (s128 >= s64x128 && s128 >= s64) return launch_cfg<2, 2, 2, 2>(p, st);  // 128 x 128
      if (s64x128 >= s64) return launch_cfg<1, 2, 2, 2>(p, st);                  // 64 x 128
      return launch_cfg<1, 1, 2, 2>(p, st);                                      // 64 x 64
    }
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
