// conv1d_x3.hip -- BigVGAN Conv1d / ConvTranspose1d as an implicit GEMM on the bf16 matrix cores of gfx950, fp32-accurate
// (rows V1, V2, V3 of SURVEY.md 8(a); same operator, call sites and epilogue as conv1d.hip).
//
// An fp32 number is EXACTLY the sum of three bf16 numbers (x = h + m + l: each piece is the remainder before it rounded to
// 8 significant bits, 8 + 8 + 8 = 24).  A product x*w = (h+m+l)(h'+m'+l') is carried to 2^-24 relative by the six partial
// products hh' hm' mh' mm' hl' lh' (the three dropped ones are below the rounding of one fp32 multiply); each is exact in the
// fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Six of those replace eight v_mfma_f32_32x32x2_f32 per 16 input channels and run
// 16x as many multiply-adds per cycle: 2.7x the fp32 MFMA rate for an fp32-quality result (against fp64 the error of a 8448-term
// dot is 1.5e-6 rms relative, the library's fp32 GEMM 1.6e-6; the full generator stays inside the 2e-4 parity bound).
// `tools/spike_mfma_bf16x6.hip`: this loop 218-252 TFLOP/s fp32-equivalent where the fp32-MFMA loop of conv1d.hip does 123-135.
//
// Operands arrive already split:
//   x planes  Xp[b][plane 3][C/8][T] 16-byte units = eight consecutive channels of one time step (written by the Snake pass /
//             the split pass): the B operand of a lane is ONE unit, the x tile of a chunk is copied by 16-byte LDS-DMA
//   weights   Wx[phase*tap][Cin/16][plane 3][k-half 2][Cout_pad] 16-byte units (eight consecutive input channels of one output
//             channel), streamed from L2 one tap ahead as in conv1d.hip.
#include <algorithm>

#include "conv.h"

namespace ixtts {

typedef conv_f32x16 cx_f32x16;
typedef __bf16 cx_bf16x8 __attribute__((ext_vector_type(8)));

template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv1d_x3_kernel(ConvParams p) {
  constexpr int BM = 32 * MT * WM;
  constexpr int BN = 32 * NT * WN;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  extern __shared__ uint4 Xq[];  // 2 buffers x [plane 3][octet 4][XW] units

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

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  cx_f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int C8 = (p.Cin + 7) >> 3;           // octets of a plane
  const int ng = p.Cin_pad >> 4;             // 16-channel groups (Cin_pad is a multiple of 16 here)
  const int nchunks = (ng + 1) >> 1;         // two groups (32 channels) per chunk
  const uint4* xp = reinterpret_cast<const uint4*>(p.x) + (size_t)b * 3 * C8 * p.Tin;
  const uint4* zero = reinterpret_cast<const uint4*>(p.zeros);
  // A operand base of this lane: [.. tap][group][plane][lh][co]
  const uint4* wq = reinterpret_cast<const uint4*>(p.wp) + (size_t)phase * p.ntap * ng * 6 * p.Cout_pad + (size_t)lh * p.Cout_pad + m0 + wm * (32 * MT) + l31;
  const size_t pstride = (size_t)2 * p.Cout_pad;  // units per plane
  const size_t gstride = 3 * pstride;             // units per group
  const size_t tstride = (size_t)ng * gstride;    // units per tap
  const int lo = q0 + p.off0 - (p.dil < 0 ? span : 0);  // smallest input index any tap of this tile touches

  // ---- x tiles by 16-byte LDS-DMA into two buffers, the copy of chunk g+1 in flight under the MFMAs of chunk g.  A chunk is
  // 12 rows (3 planes x 4 octets) of XW units; a wave copies 3 rows in pieces of 64 units (the last piece of a row is shifted
  // back to end at unit XW).  Every lane supplies its own source address: time steps outside [0, Tin) and octets beyond the
  // tensor read the zero page.
  const int np = (XW + 63) >> 6;
  auto issue_dma = [&](int g, uint4* __restrict__ dst) {
    const int o0 = min(g, nchunks - 1) * 4;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const int row = wave * 3 + rr;          // plane = row / 4, octet of the chunk = row % 4
      const int pl = row >> 2, oc = o0 + (row & 3);
      const uint4* srow = xp + ((size_t)pl * C8 + min(oc, C8 - 1)) * p.Tin;
      uint4* drow = dst + row * XW;
      for (int pc = 0; pc < np; ++pc) {
        const int cs = (pc < np - 1) ? pc * 64 : XW - 64;
        const int t = lo + cs + lane;
        const uint4* src = (t >= 0 && t < p.Tin && oc < C8) ? srow + t : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(drow + cs), 16, 0, 0);
      }
    }
  };
  // A fragments of (chunk g, tap): [group of the chunk][plane][row tile]; unconditional loads, indices clamped
  auto load_a = [&](uint4 (&a)[2][3][MT], int g, int tap) {
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < MT; ++i) a[gg][pl][i] = wq[(size_t)tap * tstride + (size_t)min(g * 2 + gg, ng - 1) * gstride + pl * pstride + i * 32];
  };
  auto load_b = [&](uint4 (&bq)[3][NT], const uint4* xrow, int gg) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int j = 0; j < NT; ++j) bq[pl][j] = xrow[(pl * 4 + gg * 2) * XW + j * 32];
  };
  const bool wave_live = m0 + wm * (32 * MT) < p.Cout;
  uint4 a_cur[2][3][MT], a_nxt[2][3][MT];
  // B fragments: group gg of a tap lives in bq[gg]; while group 0 runs, group 1 is read into bq[1]; while group 1 runs, group 0
  // of the NEXT tap is read into bq[0] (the LDS reads of a group always go out one group of MFMAs ahead)
  uint4 bq[2][3][NT];
  auto run_tap = [&](const uint4* Xc, int g, int tap) {
    const bool last_tap = tap + 1 >= p.ntap;
    load_a(a_nxt, last_tap ? min(g + 1, nchunks - 1) : g, last_tap ? 0 : tap + 1);
    const int xoff = (p.dil >= 0) ? tap * adil : (p.ntap - 1 - tap) * adil;
    const int tap_n = last_tap ? tap : tap + 1;  // (past the last tap of a chunk the prefetched fragments are not used)
    const int xoff_n = (p.dil >= 0) ? tap_n * adil : (p.ntap - 1 - tap_n) * adil;
    const uint4* xrow = Xc + lh * XW + wn * (32 * NT) + l31 + xoff;
    const uint4* xrow_n = Xc + lh * XW + wn * (32 * NT) + l31 + xoff_n;
    if (tap == 0) load_b(bq[0], xrow, 0);
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      if (gg == 0) load_b(bq[1], xrow, 1);
      else load_b(bq[0], xrow_n, 0);
      __builtin_amdgcn_sched_barrier(0);  // the next group's LDS reads go out before this group's MFMAs
      if (g * 2 + gg < ng && wave_live) {  // (wave-uniform) a ragged last chunk has one group; a wave whose rows are all padding idles
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const cx_bf16x8 ah = __builtin_bit_cast(cx_bf16x8, a_cur[gg][0][i]), am = __builtin_bit_cast(cx_bf16x8, a_cur[gg][1][i]),
                            al = __builtin_bit_cast(cx_bf16x8, a_cur[gg][2][i]);
            const cx_bf16x8 bh = __builtin_bit_cast(cx_bf16x8, bq[gg][0][j]), bm = __builtin_bit_cast(cx_bf16x8, bq[gg][1][j]),
                            bl = __builtin_bit_cast(cx_bf16x8, bq[gg][2][j]);
            // smallest partial products first
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
          }
      }
    }
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < MT; ++i) a_cur[gg][pl][i] = a_nxt[gg][pl][i];
  };

  issue_dma(0, Xq);
  load_a(a_cur, 0, 0);
  auto chunk_step = [&](int g, const uint4* __restrict__ cur, uint4* __restrict__ nxt) {
    // own copies of chunk g have landed (they are older than the 6*MT A loads still in flight), then everybody's.  With a single
    // tap the copy of chunk g was issued AFTER the chunk's only A loads (run_tap(.., 0) precedes issue_dma), so it is the
    // youngest thing in flight: wait for everything (a transposed conv with kernel == stride; the shipped generator has none)
    if (p.ntap >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * MT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    run_tap(cur, g, 0);
    if (g + 1 < nchunks) issue_dma(g + 1, nxt);  // the other buffer was last read in chunk g-1, which every wave left before the barrier above
    for (int tap = 1; tap < p.ntap; ++tap) run_tap(cur, g, tap);
  };
  const int bufsz = 12 * XW;  // (a conv of at most 32 input channels has one chunk and is given one buffer: twice the workgroups per CU)
  for (int g = 0; g < nchunks; ++g) chunk_step(g, Xq + (g & 1) * bufsz, Xq + ((g + 1) & 1) * bufsz);

  conv_epilogue<MT, NT>(p, acc, m0 + wm * (32 * MT), q0 + wn * (32 * NT), b, phase, l31, lh);
}

template <int MT, int NT, int WM, int WN>
static int launch_x3_cfg(const ConvParams& p0, hipStream_t st) {
  ConvParams p = p0;
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  p.m_tiles = ceil_div(p.Cout, BM);
  p.n_tiles = ceil_div(p.Nq, BN);
  IX_ARG(p.Cout_pad % BM == 0 && p.Cout_pad >= p.m_tiles * BM, "conv_x3: Cout_pad %d not a multiple of BM %d", p.Cout_pad, BM);
  const int adil = p.dil < 0 ? -p.dil : p.dil;
  const int XW = BN + (p.ntap - 1) * adil;
  const size_t smem = (size_t)(p.Cin_pad > 32 ? 2 : 1) * 12 * XW * 16;  // two x-tile buffers of 12 rows (one when there is a single chunk)
  IX_ARG(smem <= 160 * 1024, "conv_x3: LDS tile %zu B too large", smem);
  auto kern = conv1d_x3_kernel<MT, NT, WM, WN>;
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

// Cout padding of the packed weights: a whole number of the row tiles that may be used for this Cout
int conv_x3_cout_pad(int Cout) { return Cout <= 32 ? 32 : Cout <= 64 ? 64 : (Cout + 127) / 128 * 128; }

// Workgroups of one launch run in rounds over the CUs' slots (co-resident workgroups of a CU share its matrix pipe); a
// workgroup's LDS (two x-tile buffers) decides how many fit a CU.  Score = useful fraction of the MFMA work of those rounds
// (tile quantisation in both directions) x a mild preference for the bigger tile (fewer L2 bytes per flop).
static double x3_tile_score(const ConvParams& p, int BM, int BN, int max_occ, double pref) {
  const int adil = p.dil < 0 ? -p.dil : p.dil;
  const size_t smem = (size_t)(p.Cin_pad > 32 ? 2 : 1) * 12 * (BN + (p.ntap - 1) * adil) * 16;
  const int occ = std::max(1, std::min(max_occ, (int)((160 * 1024) / (smem + 512))));
  const double wgs = (double)ceil_div(p.Cout, BM) * ceil_div(p.Nq, BN) * p.B * p.nphase;
  const double slots = 256.0 * occ;
  const double rounds = (double)(((long long)wgs + (long long)slots - 1) / (long long)slots);
  const double useful = (double)p.Cout * p.Nq * p.B * p.nphase / (wgs * BM * BN);
  return useful * wgs / (rounds * slots) * pref;
}

// p.x = the x planes, p.wp = the weight planes (see the header); everything else as launch_conv1d.
// Wave layouts: every wave owns 32 output channels (one A fragment set per 16 input channels) and as many columns as the tile
// allows -- 128 in the 128-row tile (4 x 1 waves), 64 in the 64-row tile (2 x 2): the A operands come from L2, and L2 -> CU
// bandwidth is what bounds this loop (the spike loses a third of its rate when they are added), so a wave re-uses each of them
// for as many column tiles as its registers hold.
int launch_conv1d_x3(const ConvParams& p, hipStream_t st) {
  IX_ARG(p.Cin_pad % 16 == 0, "conv_x3: Cin_pad %d not a multiple of 16", p.Cin_pad);
  if (p.Cout_pad == 32) return launch_x3_cfg<1, 1, 1, 4>(p, st);  // 32 x 128
  if (p.Cout_pad == 64) return launch_x3_cfg<1, 2, 2, 2>(p, st);  // 64 x 128
  // (few taps per x tile -- the polyphase transposed convs have 2 to 4 -- and the tile copy weighs as much as the weights: the
  // 128-row tile halves both per flop: 252 against 369 us for the 768 -> 384 up-sampling conv)
  const double s128 = x3_tile_score(p, 128, 128, 2, 1.00), s96 = x3_tile_score(p, 128, 96, 2, 0.96),
               s64 = x3_tile_score(p, 64, 128, 3, p.ntap >= 3 ? 0.93 : 0.65);
  if (s128 >= s64 && s128 >= s96) return launch_x3_cfg<1, 4, 4, 1>(p, st);  // 128 x 128
  if (s96 >= s64) return launch_x3_cfg<1, 3, 4, 1>(p, st);                   // 128 x 96: e.g. 6 x 79 tiles on 512 slots where 6 x 60 leave 30 % idle
  return launch_x3_cfg<1, 2, 2, 2>(p, st);                                   // 64 x 128
}

// ------------------------------------------------------------------------------------
// fp32 [B][C][T] -> x planes [B][3][C8][T][8] bf16 (the inputs that do not come from a Snake pass: the mel, the stage outputs
// the transposed convs read).  One (octet, time step) per thread: eight strided 4-byte reads (coalesced across the wave),
// three 16-byte writes.

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, uint4* __restrict__ xp, int C, int T) {
  const int C8 = (C + 7) >> 3;
  const int t = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float* xb = x + (size_t)b * C * T;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (o * 8 + e < C) ? xb[(size_t)(o * 8 + e) * T + t] : 0.f;
  uint4 ph, pm, pl;
  split8_bf16x3(v, ph, pm, pl);
  uint4* dst = xp + ((size_t)b * 3 * C8 + o) * T + t;
  dst[0] = ph;
  dst[(size_t)C8 * T] = pm;
  dst[(size_t)2 * C8 * T] = pl;
}

int launch_split_planes(const float* x, void* xp, int B, int C, int T, hipStream_t st) {
  if (B * C == 0 || T == 0) return IXTTS_OK;
  IX_ARG((C + 7) / 8 <= 65535 && B <= 65535, "split_planes: grid too large");
  hipLaunchKernelGGL(split_planes_kernel, dim3(ceil_div(T, 256), (C + 7) / 8, B), dim3(256), 0, st, x, reinterpret_cast<uint4*>(xp), C, T);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

}  // namespace ixtts
