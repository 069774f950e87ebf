// gemm_x6.hip -- fp32-accurate row-major GEMM  C[M][N] (+)= A[M][K] . W[N][K]^T + bias  on the bf16 matrix cores of gfx950:
// the linear layers of the s2mel DiT / WaveNet (row N1 of SURVEY.md 8(f)): `Attention.wqkv / wo`, `FeedForward.w1|w3 / w2`
// (indextts/s2mel/modules/gpt_fast/model.py:242-326), the WaveNet's k = 5 convs as five row-shifted GEMMs and its 1x1
// res / skip convs (wavenet.py:103-174), the AdaLN projections and the merge / skip linears (diffusion_transformer.py:186-257).
//
// Arithmetic: every fp32 product as six exact bf16 partial products with fp32 accumulation (conv1d_x3.hip's scheme):
// x = h + m + l, each piece the remainder before it rounded to 8 significant bits; hh' hm' mh' mm' hl' lh' carry x.w to 2^-24.
//
// What the r02 attempt (128 x 128 tiles, operand fragments straight from L2) lost on was operand bytes per MFMA.  Here both operands
// go through LDS, shared by the four waves of a 256 x 128 / 128 x 128 tile:
//   * the WEIGHTS are split once at load (`ixtts_gemm_x6_pack`) into planes [k/16][plane 3][k-half 2][N_pad] of 16-byte units
//     (eight consecutive k of one output feature = the B operand of a lane);
//   * the ACTIVATIONS are split into the same layout [k/16][3][2][M_pad] by a small pass (`ixtts_gemm_x6_split`: one coalesced read of
//     the fp32 rows, 6 B per element written) -- a producer that writes the planes itself saves that pass;
//   * both are copied global -> LDS by LDS-DMA into a 4-stage ring, THREE 16-deep steps ahead of the MFMAs: with one step of cover
//     (the first version) every barrier waited out a full L2 / Infinity-Cache round trip and the kernel ran at the library's speed;
//   * the blockIdx -> tile map gives each XCD a contiguous run of tiles, row-tile major: the tiles of one XCD share their A rows in
//     its L2.
// MFMA operands: A = activations (lane: row l & 31, k-half l >> 5), B = weights (lane: output feature l & 31, k-half l >> 5), so an
// accumulator register holds 32 CONSECUTIVE output features of one row: the epilogue's stores are 128-byte row segments.
#include "common.h"

namespace ixtts {

typedef float gx_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 gx_bf16x8 __attribute__((ext_vector_type(8)));
typedef float gx_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gx_u32x4 __attribute__((ext_vector_type(4)));


// fp32 [N][K] -> packed planes [K/16][3][2][N_pad] units; rows n >= N are zero
__global__ void gemm_x6_pack_kernel(const float* __restrict__ w, uint4* __restrict__ out, int N, int K, int Npad) {
  const long total = (long)(K / 8) * Npad;  // one thread per (k-octet, n)
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int n = (int)(idx % Npad);
    const int ko = (int)(idx / Npad);  // k-octet: g = ko / 2, kh = ko % 2
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = n < N ? w[(long)n * K + ko * 8 + e] : 0.f;
    uint4 ph, pm, pl;
    split8_bf16x3(v, ph, pm, pl);
    const int g = ko >> 1, kh = ko & 1;
    uint4* base = out + ((long)g * 6) * Npad + n;
    base[(long)(0 * 2 + kh) * Npad] = ph;
    base[(long)(1 * 2 + kh) * Npad] = pm;
    base[(long)(2 * 2 + kh) * Npad] = pl;
  }
}

// Activations A [M][K] (row stride lda) -> the same packed planes [K/16][3][2][Mpad] units, rows >= M zero.  One workgroup = 64 rows x
// 64 k: coalesced 256-byte row reads into an LDS tile, then one (row, k-octet) unit per thread and plane, rows contiguous across lanes.
__global__ __launch_bounds__(256) void gemm_x6_split_kernel(const float* __restrict__ a, long lda, uint4* __restrict__ out, int M, int K, long Mpad) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (t >> 4) + 16 * i, kq = t & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < M) v = *reinterpret_cast<const float4*>(a + (long)(r0 + r) * lda + k0 + 4 * kq);
    tile[r][4 * kq + 0] = v.x; tile[r][4 * kq + 1] = v.y; tile[r][4 * kq + 2] = v.z; tile[r][4 * kq + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = t & 63, o = (t >> 6) + 4 * u;  // k-octet of this 64-wide k block
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[r][o * 8 + e];
    uint4 ph, pm, pl;
    split8_bf16x3(v, ph, pm, pl);
    const int ko = (k0 >> 3) + o, g = ko >> 1, kh = ko & 1;
    uint4* base = out + ((long)g * 6) * Mpad + r0 + r;
    base[(long)(0 * 2 + kh) * Mpad] = ph;
    base[(long)(1 * 2 + kh) * Mpad] = pm;
    base[(long)(2 * 2 + kh) * Mpad] = pl;
  }
}

struct GemmX6Params {
  const uint4* Ap;   // activation planes [K/16][6][Mpad], this GEMM's rows start at unit `arow0`
  long Mpad;
  long arow0;
  const uint4* Wp;   // weight planes [K/16][6][Npad]
  const float* bias;
  float* C;
  long ldc;
  int M, N, Npad, K, accumulate, m_tiles, n_tiles;
  unsigned long long* dbg;  // DBG == 3: per-step phase stamps of workgroup 0, wave 0 (developer timing)
};


// C tile BM x BN = (64 MT) x (64 NT), 4 waves as 2 x 2.  Both operands arrive as bf16 planes and are copied global -> LDS by LDS-DMA
// into a 4-stage ring, three 16-deep steps ahead; every memory operation of the main loop is issued by hand (DMA builtins, fragment
// reads from ONE asm block that ends in its own wait), so the only vmcnt waits are the counted ones below: left to the compiler, a
// ds_read of a stage that a copy of an earlier loop iteration filled drains every copy in flight (vmcnt(0)) -- it cannot count across
// the back edge.
// NST LDS stages: the operand copies run NST - 1 steps ahead of the MFMAs.  DBG (developer timing, results meaningless): 1 = no operand
// copies in the main loop, 2 = no fragment reads / MFMAs, 3 = phase stamps.
template <int MT, int NT, int NST, int DBG = 0>
__global__ __launch_bounds__(256, (NST * (MT + NT) * 6 * 64 * 16 <= 80 * 1024) ? 2 : 1) void gemm_x6_kernel(GemmX6Params p) {
  constexpr int BM = 64 * MT, BN = 64 * NT;
  constexpr int A_UNITS = 6 * BM, W_UNITS = 6 * BN, STAGE = A_UNITS + W_UNITS;
  constexpr int NDMA = (6 * (MT + NT)) / 4;       // LDS-DMA instructions per wave and step
  static_assert((6 * (MT + NT)) % 4 == 0, "the copies are dealt evenly to the 4 waves");
  constexpr int D = NST - 1;                      // steps the copies run ahead
  extern __shared__ uint4 smem[];                 // NST stages x [A image [3][2][BM] | W image [3][2][BN]]

  // ---- XCD-aware tile id: blocks b and b + 8 share an XCD; each XCD takes a contiguous run of tiles, row-tile major
  const int nwg = p.m_tiles * p.n_tiles;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int lin = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
  const int m_tile = lin / p.n_tiles, n_tile = lin - m_tile * p.n_tiles;
  const int m0 = m_tile * BM, n0 = n_tile * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  gx_f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one step's copies: 6 runs (plane, k-half) of BM + BN units, in pieces of 64 units (one LDS-DMA wave-instruction each)
  const uint4* asrc = p.Ap + p.arow0 + m0 + lane;
  const uint4* wsrc = p.Wp + n0 + lane;
  auto issue = [&](int s, uint4* st) {
#pragma unroll
    for (int q = 0; q < NDMA; ++q) {
      const int e = q * 4 + wave;  // piece id among the step's 6 (MT + NT)
      if (e < 6 * MT) {
        const int run = e / MT, piece = e % MT;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + ((long)s * 6 + run) * p.Mpad + piece * 64),
                                         (__attribute__((address_space(3))) void*)(st + run * BM + piece * 64), 16, 0, 0);
      } else {
        const int e2 = e - 6 * MT, run = e2 / NT, piece = e2 % NT;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + ((long)s * 6 + run) * p.Npad + piece * 64),
                                         (__attribute__((address_space(3))) void*)(st + A_UNITS + run * BN + piece * 64), 16, 0, 0);
      }
    }
  };

  const int steps = p.K >> 4;
  auto wg_stamp = [&](int k) {  // DBG == 3: per-workgroup wall clock (100 MHz) at entry / loop start / loop end / exit, behind the step stamps
    if constexpr (DBG == 3) {
      if (tid == 0) p.dbg[128 + (long)blockIdx.x * 4 + k] = __builtin_amdgcn_s_memrealtime();
    }
  };
  // counted wait: leave the copies of the `y` youngest steps in flight (vmcnt counts in order)
  auto wait_younger = [&](int y) {
    if (y >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
    else if (y == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(D >= 1 && D <= 3, "1..3 steps of copies in flight behind the one waited for");
  wg_stamp(0);
  for (int t = 0; t < D && t < steps; ++t) issue(t, smem + t * STAGE);
  wait_younger(min(D, steps) - 1);  // step 0's copies (the oldest) have landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  wg_stamp(1);

  // Step s computes on stage s % NST and, first, issues the copies of step s + D into stage (s + D) % NST (last read in step s - 1,
  // which every wave has left).  Before its closing barrier a wave waits until its share of step s + 1's copies has landed; the
  // copies of the steps behind that stay in flight.  Every memory operation here is issued by hand, so a run-time stage index and
  // run-time issue condition are fine: the compiler has no wait of its own to miscount.
  const unsigned base_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)smem;
  int st_cur = 0, st_nxt = D % NST;
  for (int s = 0; s < steps; ++s) {
    auto stamp = [&](int k) {
      if constexpr (DBG == 3) {
        if (blockIdx.x == 0 && tid == 0 && s < 16) {
          p.dbg[s * 8 + k] = __builtin_amdgcn_s_memtime();
          if (k == 0) p.dbg[s * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        }
      }
    };
    stamp(0);
    if (DBG != 1 && s + D < steps) issue(s + D, smem + st_nxt * STAGE);
    stamp(1);
    const unsigned cur_addr = base_addr + (unsigned)(st_cur * STAGE * 16);
    if constexpr (DBG != 2) {
      // fragments: A [plane][i] at unit (plane * 2 + lh) * BM + wm * 32 MT + i * 32 + l31; W likewise behind the A image
      gx_u32x4 af[3][MT], bf[3][NT];
      const unsigned aaddr = cur_addr + (unsigned)((lh * BM + wm * (32 * MT) + l31) * 16);
      const unsigned waddr = cur_addr + (unsigned)((A_UNITS + lh * BN + wn * (32 * NT) + l31) * 16);
      if constexpr (MT == 4 && NT == 2) {
        asm volatile(
            "ds_read_b128 %0, %18 offset:%20\n ds_read_b128 %1, %18 offset:%21\n ds_read_b128 %2, %18 offset:%22\n ds_read_b128 %3, %18 offset:%23\n"
            "ds_read_b128 %4, %18 offset:%24\n ds_read_b128 %5, %18 offset:%25\n ds_read_b128 %6, %18 offset:%26\n ds_read_b128 %7, %18 offset:%27\n"
            "ds_read_b128 %8, %18 offset:%28\n ds_read_b128 %9, %18 offset:%29\n ds_read_b128 %10, %18 offset:%30\n ds_read_b128 %11, %18 offset:%31\n"
            "ds_read_b128 %12, %19 offset:%32\n ds_read_b128 %13, %19 offset:%33\n ds_read_b128 %14, %19 offset:%34\n ds_read_b128 %15, %19 offset:%35\n"
            "ds_read_b128 %16, %19 offset:%36\n ds_read_b128 %17, %19 offset:%37\n s_waitcnt lgkmcnt(0)"
            : "=&v"(af[0][0]), "=&v"(af[0][1]), "=&v"(af[0][2]), "=&v"(af[0][3]), "=&v"(af[1][0]), "=&v"(af[1][1]), "=&v"(af[1][2]), "=&v"(af[1][3]),
              "=&v"(af[2][0]), "=&v"(af[2][1]), "=&v"(af[2][2]), "=&v"(af[2][3]), "=&v"(bf[0][0]), "=&v"(bf[0][1]), "=&v"(bf[1][0]), "=&v"(bf[1][1]),
              "=&v"(bf[2][0]), "=&v"(bf[2][1])
            : "v"(aaddr), "v"(waddr), "n"((0 * 2 * BM + 0) * 16), "n"((0 * 2 * BM + 32) * 16), "n"((0 * 2 * BM + 64) * 16), "n"((0 * 2 * BM + 96) * 16),
              "n"((1 * 2 * BM + 0) * 16), "n"((1 * 2 * BM + 32) * 16), "n"((1 * 2 * BM + 64) * 16), "n"((1 * 2 * BM + 96) * 16), "n"((2 * 2 * BM + 0) * 16),
              "n"((2 * 2 * BM + 32) * 16), "n"((2 * 2 * BM + 64) * 16), "n"((2 * 2 * BM + 96) * 16), "n"((0 * 2 * BN + 0) * 16), "n"((0 * 2 * BN + 32) * 16),
              "n"((1 * 2 * BN + 0) * 16), "n"((1 * 2 * BN + 32) * 16), "n"((2 * 2 * BN + 0) * 16), "n"((2 * 2 * BN + 32) * 16)
            : "memory");
      } else {
        static_assert(MT == 2 && NT == 2, "fragment block written for 256 x 128 and 128 x 128 tiles");
        asm volatile(
            "ds_read_b128 %0, %12 offset:%14\n ds_read_b128 %1, %12 offset:%15\n ds_read_b128 %2, %12 offset:%16\n ds_read_b128 %3, %12 offset:%17\n"
            "ds_read_b128 %4, %12 offset:%18\n ds_read_b128 %5, %12 offset:%19\n ds_read_b128 %6, %13 offset:%20\n ds_read_b128 %7, %13 offset:%21\n"
            "ds_read_b128 %8, %13 offset:%22\n ds_read_b128 %9, %13 offset:%23\n ds_read_b128 %10, %13 offset:%24\n ds_read_b128 %11, %13 offset:%25\n"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(af[0][0]), "=&v"(af[0][1]), "=&v"(af[1][0]), "=&v"(af[1][1]), "=&v"(af[2][0]), "=&v"(af[2][1]), "=&v"(bf[0][0]), "=&v"(bf[0][1]),
              "=&v"(bf[1][0]), "=&v"(bf[1][1]), "=&v"(bf[2][0]), "=&v"(bf[2][1])
            : "v"(aaddr), "v"(waddr), "n"((0 * 2 * BM + 0) * 16), "n"((0 * 2 * BM + 32) * 16), "n"((1 * 2 * BM + 0) * 16), "n"((1 * 2 * BM + 32) * 16),
              "n"((2 * 2 * BM + 0) * 16), "n"((2 * 2 * BM + 32) * 16), "n"((0 * 2 * BN + 0) * 16), "n"((0 * 2 * BN + 32) * 16), "n"((1 * 2 * BN + 0) * 16),
              "n"((1 * 2 * BN + 32) * 16), "n"((2 * 2 * BN + 0) * 16), "n"((2 * 2 * BN + 32) * 16)
            : "memory");
      }
      stamp(2);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const gx_bf16x8 ah = __builtin_bit_cast(gx_bf16x8, af[0][i]), am = __builtin_bit_cast(gx_bf16x8, af[1][i]), al = __builtin_bit_cast(gx_bf16x8, af[2][i]);
          const gx_bf16x8 bh = __builtin_bit_cast(gx_bf16x8, bf[0][j]), bm = __builtin_bit_cast(gx_bf16x8, bf[1][j]), bl = __builtin_bit_cast(gx_bf16x8, bf[2][j]);
          // smallest partial products first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
        }
      if constexpr (DBG == 3) asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[MT - 1][NT - 1][15]));  // (the stamp below sits behind the last MFMAs' results)
    }
    stamp(3);
    wait_younger(DBG == 1 ? 0 : max(0, min(D - 1, steps - s - 2)));
    stamp(4);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(5);
    st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
    st_nxt = st_nxt + 1 == NST ? 0 : st_nxt + 1;
  }

  wg_stamp(2);
  // ---- epilogue: + bias (+ what C held): register r of a lane = row (r & 3) + 8 (r >> 2) + 4 lh of the 32 x 32 tile, column l31
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * (32 * NT) + j * 32 + l31;
    const bool col_ok = col < p.N;
    const float bv = (p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int rbase = m0 + wm * (32 * MT) + i * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (col_ok && row < p.M) {
          float* c = p.C + (long)row * p.ldc + col;
          float v = acc[i][j][r] + bv;
          if (p.accumulate) v += *c;
          *c = v;
        }
      }
    }
  }
  if constexpr (DBG == 3) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_stamp(3);
  }
}

template <int MT, int NT, int NST, int DBG = 0>
static int launch_gemm_x6(GemmX6Params p, hipStream_t st) {
  constexpr int BM = 64 * MT, BN = 64 * NT;
  constexpr size_t smem = (size_t)NST * (6 * BM + 6 * BN) * 16;
  static_assert(smem <= 160 * 1024, "LDS stages of this tile shape");
  p.m_tiles = ceil_div(p.M, BM);
  p.n_tiles = ceil_div(p.N, BN);
  IX_ARG((long)p.n_tiles * BN <= p.Npad, "gemm_x6: packed weights padded to %d features, the %d-wide tile needs %d", p.Npad, BN, p.n_tiles * BN);
  IX_ARG(p.arow0 + (long)p.m_tiles * BM <= p.Mpad, "gemm_x6: activation planes hold %ld rows, the %d-row tiles reach row %ld", p.Mpad, BM, p.arow0 + (long)p.m_tiles * BM);
  auto kern = gemm_x6_kernel<MT, NT, NST, DBG>;
  static bool done = false;
  if (!done) {
    IX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    done = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.m_tiles * p.n_tiles), dim3(256), smem, st, p);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

}  // namespace ixtts

using namespace ixtts;

extern "C" size_t ixtts_gemm_x6_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 64) return 0;
  const size_t Npad = (size_t)(N + 255) / 256 * 256;
  return (size_t)(K / 16) * 6 * Npad * 16;
}

extern "C" int ixtts_gemm_x6_pack(const float* w_dev, void* packed_dev, int N, int K, void* stream) {
  IX_ARG(w_dev && packed_dev && N > 0 && K > 0 && K % 64 == 0, "gemm_x6_pack: bad argument (K must be a multiple of 64)");
  const int Npad = (N + 255) / 256 * 256;
  const long total = (long)(K / 8) * Npad;
  hipLaunchKernelGGL(gemm_x6_pack_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, w_dev,
                     reinterpret_cast<uint4*>(packed_dev), N, K, Npad);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

extern "C" long ixtts_gemm_x6_rows_padded(long rows) { return rows <= 0 ? 0 : (rows + 255) / 256 * 256 + 256; }

extern "C" int ixtts_gemm_x6_split(const float* a_dev, long lda, void* planes_dev, long rows, int K, void* stream) {
  IX_ARG(a_dev && planes_dev && rows > 0 && K > 0 && K % 64 == 0, "gemm_x6_split: bad argument (K must be a multiple of 64)");
  IX_ARG(lda >= K && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(a_dev) & 15) == 0, "gemm_x6_split: rows must be 16-byte aligned (lda %ld)", lda);
  const long Mpad = ixtts_gemm_x6_rows_padded(rows);
  hipLaunchKernelGGL(gemm_x6_split_kernel, dim3((unsigned)(Mpad / 64), K / 64), dim3(256), 0, (hipStream_t)stream, a_dev, lda,
                     reinterpret_cast<uint4*>(planes_dev), (int)rows, K, Mpad);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

extern "C" int ixtts_gemm_x6_f32(const void* a_planes_dev, long rows_total, long row0, const void* packed_dev, const float* bias_dev, float* c_dev,
                                 long ldc, int M, int N, int K, int accumulate, int tile, void* stream) {
  IX_ARG(a_planes_dev && packed_dev && c_dev && M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_x6: bad argument (K must be a multiple of 64: the split pass works on 64-wide k blocks)");
  IX_ARG(ldc >= N && row0 >= 0 && row0 + M <= rows_total, "gemm_x6: rows [%ld, %ld) of %ld, ldc %ld", row0, row0 + M, rows_total, ldc);
  GemmX6Params p;
  p.Ap = reinterpret_cast<const uint4*>(a_planes_dev); p.Mpad = ixtts_gemm_x6_rows_padded(rows_total); p.arow0 = row0;
  p.Wp = reinterpret_cast<const uint4*>(packed_dev); p.bias = bias_dev; p.C = c_dev; p.ldc = ldc;
  p.M = M; p.N = N; p.Npad = (N + 255) / 256 * 256; p.K = K; p.accumulate = accumulate ? 1 : 0; p.m_tiles = p.n_tiles = 0;
  p.dbg = (tile == 32 || tile == 33) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(bias_dev)) : nullptr;  // developer stamps travel in the bias slot
  if (p.dbg) p.bias = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (tile == 0) {  // by fill of the 256 CUs: the wider tile while it still gives most CUs work (operand bytes per MFMA fall with the tile)
    const long t42 = (long)ceil_div(M, 256) * ceil_div(N, 128);
    tile = t42 >= 140 ? 2 : 3;
  }
  switch (tile) {
    // two workgroups per CU (2 x 73.7 KB of LDS): one's copies, fragment reads and epilogue run under the other's MFMAs
    case 2: return launch_gemm_x6<4, 2, 2>(p, st);
    case 3: return launch_gemm_x6<2, 2, 3>(p, st);
    // one workgroup per CU, copies three steps ahead (A/B timing)
    case 4: return launch_gemm_x6<4, 2, 4>(p, st);
    case 5: return launch_gemm_x6<2, 2, 4>(p, st);
    // developer timing variants (results meaningless)
    case 12: return launch_gemm_x6<4, 2, 2, 1>(p, st);
    case 22: return launch_gemm_x6<4, 2, 2, 2>(p, st);
    case 32: return launch_gemm_x6<4, 2, 2, 3>(p, st);
    case 33: return launch_gemm_x6<2, 2, 3, 3>(p, st);
  }
  set_error("gemm_x6: tile %d (0 auto; 2 / 3: 256x128 / 128x128, two workgroups per CU; 4 / 5: the same, one per CU with a 4-stage ring)", tile);
  return IXTTS_ERR_ARG;
}
