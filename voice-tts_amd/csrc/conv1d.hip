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

typedef conv_f32x16 f32x16;

constexpr int CG = 8;   // input channels per weight group (one float4 per lane = 4 MFMA k-steps)
// CKG groups per K-chunk (template parameter): the x tile in LDS covers 8 CKG input channels -- 32, or 24 for the 24- / 48-channel
// stages, whose last 32-channel chunk would run a quarter of its MFMAs on zeroed weights

// Weights are packed Wq[phase][tap][ci/8][ci%2][co][(ci%8)/2]: lane (co = l31, k-parity = lh) reads ONE float4
// holding its A operand for the 4 consecutive k-steps of a channel group; a half-wave reads 512 contiguous bytes.
// They stream from L2 (the XCD-aware tile map keeps one co-slice per XCD) straight into registers, one tap ahead;
// only the x tile (+dilation halo) lives in LDS.
template <int MT, int NT, int WM, int WN, int CKG = 4, bool RAGGED = true>
__global__ __launch_bounds__(256, (MT == 1 ? 3 : 2)) void conv1d_mfma_kernel(ConvParams p) {
  static_assert(CKG == 3 || CKG == 4, "three or four 8-channel groups per chunk");
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

  // ---- x tiles: LDS-DMA (global_load_lds_dword: no VGPRs, no ds_write pass) into two buffers, the copy of chunk g+1 in
  // flight under the MFMAs of chunk g.  A chunk is 32 channel rows x XW columns; a wave copies 8 rows in pieces of 64
  // columns (the last piece of a row is shifted back to end at column XW: it rewrites a few columns with the same
  // values instead of spilling into the next row).  Every lane supplies its own source address, so padding (t outside
  // [0,Tin), channels beyond Cin) reads the zero page -- no branches, and all 32 rows are always staged, which lets the
  // ragged last chunk run the same MFMA sequence with zeroed weights.
  const int nchunks = (ngroups + CKG - 1) / CKG;
  const int np = (XW + 63) >> 6;
  auto issue_dma = [&](int g, float* __restrict__ dst) {
    const int c0 = min(g, nchunks - 1) * (CKG * CG) + wave * 8;
    float* dst0 = dst + (wave * 8) * XWP;
    if (CKG < 4 && wave >= CKG) return;  // a 24-row chunk is copied by three of the four waves
    for (int pc = 0; pc < np; ++pc) {
      const int cs = (pc < np - 1) ? pc * 64 : XW - 64;
      const int t = lo + cs + lane;
      const bool tvalid = t >= 0 && t < p.Tin;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int c = c0 + r;
        const float* src = (tvalid && c < p.Cin) ? xb + (size_t)c * p.Tin + t : p.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst0 + r * XWP + cs), 4, 0, 0);
      }
    }
  };
  // A fragments of (chunk g, tap): unconditional loads, indices clamped; groups beyond the matrix are zeroed when used
  auto load_a = [&](float4 (&a)[CKG][MT], int g, int tap) {
#pragma unroll
    for (int gg = 0; gg < CKG; ++gg)
#pragma unroll
      for (int i = 0; i < MT; ++i) a[gg][i] = wq[(size_t)tap * tstride + (size_t)min(g * CKG + gg, ngroups - 1) * gstride + i * 32];
  };
  auto load_b = [&](float (&bq)[4][NT], const float* xrow, int gg) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < NT; ++j) bq[kk][j] = xrow[(gg * CG + 2 * kk) * XWP + j * 32];
  };
  float4 a_cur[CKG][MT], a_nxt[CKG][MT];
  // one tap of one chunk: B fragments of the next group (and of the next tap's first group) are read while this group's
  // 16 MFMAs run; A fragments of the next (chunk, tap) are loaded at entry
  float bcarry[4][NT];  // B fragments of the next tap's first group, read under the current tap's last group
  auto run_tap = [&](const float* Xc, int g, int tap) {
    const bool last_tap = tap + 1 >= p.ntap;
    load_a(a_nxt, last_tap ? min(g + 1, nchunks - 1) : g, last_tap ? 0 : tap + 1);
    const int xoff = (p.dil >= 0) ? tap * adil : (p.ntap - 1 - tap) * adil;
    const int tap_n = last_tap ? tap : tap + 1;  // (past the last tap of a chunk the carried fragments are not used)
    const int xoff_n = (p.dil >= 0) ? tap_n * adil : (p.ntap - 1 - tap_n) * adil;
    const float* xrow = Xc + lh * XWP + wn * (32 * NT) + l31 + xoff;
    const float* xrow_n = Xc + lh * XWP + wn * (32 * NT) + l31 + xoff_n;
    float bq[2][4][NT];
    if (tap == 0) {
      load_b(bq[0], xrow, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < NT; ++j) bq[0][kk][j] = bcarry[kk][j];
    }
#pragma unroll
    for (int gg = 0; gg < CKG; ++gg) {
      if (gg + 1 < CKG) load_b(bq[(gg + 1) & 1], xrow, gg + 1);
      else load_b(bcarry, xrow_n, 0);
      __builtin_amdgcn_sched_barrier(0);  // the next group's LDS reads go out before this group's MFMAs
      const bool live = g * CKG + gg < ngroups;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          float av = kk == 0 ? a_cur[gg][i].x : kk == 1 ? a_cur[gg][i].y : kk == 2 ? a_cur[gg][i].z : a_cur[gg][i].w;
          if constexpr (RAGGED) av = live ? av : 0.f;  // (Cin a whole number of chunks: no select + hazard nop between the MFMAs)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[gg & 1][kk][j], acc[i][j], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int gg = 0; gg < CKG; ++gg)
#pragma unroll
      for (int i = 0; i < MT; ++i) a_cur[gg][i] = a_nxt[gg][i];
  };

  issue_dma(0, Xs);
  load_a(a_cur, 0, 0);
  // (`cur` / `nxt` restrict-qualified: the compiler then knows the tile being read is not the one the copies in flight
  // write; without that it puts a vmcnt(0) -- a wait for the NEXT chunk -- in front of the LDS reads that follow the issue)
  auto chunk_step = [&](int g, const float* __restrict__ cur, float* __restrict__ nxt) {
    // own copies of chunk g have landed (they are older than the CKG*MT A loads still in flight), then everybody's
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CKG * MT) : "memory");
    __builtin_amdgcn_s_barrier();
    run_tap(cur, g, 0);
    // the other buffer was last read in chunk g-1, which every wave left before the barrier above.  Issued after the
    // first tap so that waiting for that tap's A prefetch (older, in-order counter) does not wait for these copies
    issue_dma(g + 1, nxt);
    for (int tap = 1; tap < p.ntap; ++tap) run_tap(cur, g, tap);
  };
  const int bufsz = (CKG * CG) * XWP;
  for (int g = 0; g < nchunks; ++g) chunk_step(g, Xs + (g & 1) * bufsz, Xs + ((g + 1) & 1) * bufsz);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus copy of the last iteration, before the workgroup's LDS is released

  conv_epilogue<MT, NT>(p, acc, m0 + wm * (32 * MT), q0 + wn * (32 * NT), b, phase, l31, lh);
}

template <int MT, int NT, int WM, int WN, int CKG = 4, bool RAGGED = true>
static int launch_cfg_r(const ConvParams& p0, hipStream_t st);

template <int MT, int NT, int WM, int WN, int CKG = 4>
static int launch_cfg(const ConvParams& p0, hipStream_t st) {
  return (p0.Cin_pad / CG) % CKG == 0 ? launch_cfg_r<MT, NT, WM, WN, CKG, false>(p0, st) : launch_cfg_r<MT, NT, WM, WN, CKG, true>(p0, st);
}

template <int MT, int NT, int WM, int WN, int CKG, bool RAGGED>
static int launch_cfg_r(const ConvParams& p0, hipStream_t st) {
  ConvParams p = p0;
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  p.m_tiles = ceil_div(p.Cout, BM);
  p.n_tiles = ceil_div(p.Nq, BN);
  IX_ARG(p.Cout_pad % BM == 0 && p.Cout_pad >= p.m_tiles * BM, "conv: Cout_pad %d not a multiple of BM %d", p.Cout_pad, BM);
  int adil = p.dil < 0 ? -p.dil : p.dil;
  int XWP = (BN + (p.ntap - 1) * adil) | 1;
  size_t smem = (size_t)(2 * CKG * CG * XWP) * sizeof(float);  // two x-tile buffers
  IX_ARG(smem <= 160 * 1024, "conv: LDS tile %zu B too large", smem);
  IX_ARG(BN + (p.ntap - 1) * adil <= 512, "conv: x tile of %d columns exceeds the staging bound (512)", BN + (p.ntap - 1) * adil);
  auto kern = conv1d_mfma_kernel<MT, NT, WM, WN, CKG, RAGGED>;
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
  const bool k24 = p.Cin_pad % 24 == 0 && p.Cin_pad % 32 != 0;  // 24 / 48 input channels: 24-channel chunks, no dead group
  switch (conv_tile_bm(p.Cout)) {
    case 32: return k24 ? launch_cfg<1, 2, 1, 4, 3>(p, st) : launch_cfg<1, 2, 1, 4>(p, st);   // 32 x 256
    case 64: return k24 ? launch_cfg<2, 1, 1, 4, 3>(p, st) : launch_cfg<2, 1, 1, 4>(p, st);   // 64 x 128
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
