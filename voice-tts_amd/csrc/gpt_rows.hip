// gpt_rows.hip -- batched ("rows") causal pass of the GPT-2 trunk on gfx950: prefill of the
// prompt (row G1, model_v2.py:144-155) and the latent forward (row G9, model_v2.py:554-596).
//
// T rows go through the 24 layers together: LayerNorm rows -> MFMA GEMM against the same
// transposed/folded weight arena the decode GEMVs stream (QKV with KV-cache scatter, out-proj
// + residual, FC + gelu_new, MLP-out + residual) and a causal row-attention that reads the
// cache.  fp32 mode uses v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains, parity mode), bf16
// mode converts the activation tile to bf16 in the LDS staging pass and uses
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// Reference arithmetic: indextts/gpt/transformers_gpt2.py:480-667.
#include "gpt_engine.h"
#include "gpt_kernels.h"

namespace ixtts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// two fp32 -> packed bf16 pair (round-to-nearest-even, the plain cast lowers to v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned int pack_bf16x2(float a, float b) {
  const unsigned short lo = __builtin_bit_cast(unsigned short, (__bf16)a);
  const unsigned short hi = __builtin_bit_cast(unsigned short, (__bf16)b);
  return (unsigned int)lo | ((unsigned int)hi << 16);
}

// ---- (x - mean) * rstd per row (gain/bias are folded into the next matrix); one wave per row
template <int K, typename OT = float>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, OT* __restrict__ y, int T) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= T) return;
  constexpr int PL = K / 64;  // elements per lane (20 for 1280, 2 for 128)
  const float* xr = x + (size_t)row * K;
  float v[PL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    v[i] = xr[lane + 64 * i];
    s += v[i];
  }
  const float mean = wave_sum(s) * (1.0f / K);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const float d = v[i] - mean;
    q = fmaf(d, d, q);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / K) + 1e-5f);
  OT* yr = y + (size_t)row * K;
#pragma unroll
  for (int i = 0; i < PL; ++i) store_kv(yr + lane + 64 * i, (v[i] - mean) * rstd);
}

// ---- ln_f (explicit affine) then final_norm (explicit affine) per row -> latent rows
template <int K>
__global__ __launch_bounds__(256) void final_norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int T,
                                                               const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ w2, const float* __restrict__ b2) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= T) return;
  constexpr int PL = K / 64;
  const float* xr = x + (size_t)row * K;
  float v[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) v[i] = xr[lane + 64 * i];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const float* w = pass == 0 ? w1 : w2;
    const float* b = pass == 0 ? b1 : b2;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PL; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / K);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const float d = v[i] - mean;
      q = fmaf(d, d, q);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / K) + 1e-5f);
#pragma unroll
    for (int i = 0; i < PL; ++i) v[i] = (v[i] - mean) * rstd * w[lane + 64 * i] + b[lane + 64 * i];
  }
  float* yr = y + (size_t)row * K;
#pragma unroll
  for (int i = 0; i < PL; ++i) yr[lane + 64 * i] = v[i];
}

// ---- GEMM: C[T][N] = A[T][K] . Wt[N][K]^T + bias
enum { RE_QKV = 0, RE_RESID = 1, RE_GELU = 2 };

struct GemmArgs {
  const float* A;    // [T][K] fp32
  const void* wt;    // [N][K]
  const float* bias; // [N]
  float* out;        // RE_QKV: q [T][D]; RE_RESID: x [T][N] (+=); RE_GELU: ff [T][N]
  void* kcache;      // RE_QKV: slot+layer base [H][smax][64]
  void* vcache;
  int T, N, K, pos0, smax, D;
};

constexpr int GBN = 128;

// Tile BM x 128 x BK, 4 waves as 2 x 2, wave tile (BM/2) x 64.  bf16: BK = 64, both operands staged as bf16 in LDS (the
// fp32 activations are converted in the staging pass), rows padded to 144 B so the 16-byte fragment reads of 8 consecutive
// rows fall in 8 different bank groups; fp32: BK = 32, pitch 33.  The global loads of k-tile i+1 are issued before the
// MFMAs of k-tile i and written to LDS after them (one register set, one LDS buffer, two barriers per k-tile).
template <typename WT, int EPI, typename KVT, int BM>
__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmArgs g) {
  constexpr bool F32 = sizeof(WT) == 4;
  constexpr int BK = F32 ? 32 : 64;
  constexpr int MI = BM / 64;                              // 32-row blocks per wave
  constexpr int PITCH = F32 ? (BK + 1) : (BK + 8) / 2;     // floats per LDS row (bf16: 72 halfs = 36 floats)
  __shared__ __attribute__((aligned(16))) float As[BM * PITCH];
  __shared__ __attribute__((aligned(16))) float Ws[GBN * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * GBN;

  f32x16 acc[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.f;

  constexpr int NA = BM * BK / 4 / 256;                    // float4 of A per thread per k-tile
  constexpr int NW = F32 ? GBN * BK / 4 / 256 : GBN * BK / 8 / 256;  // 16-byte pieces of W per thread
  constexpr int AC = BK / 4;                               // float4 per A row
  constexpr int WC = F32 ? BK / 4 : BK / 8;                // 16-byte pieces per W row
  constexpr int PD = 1;  // k-tiles in flight ahead of the MFMAs (register sets); r01: 3 sets measured slower (latent pass 21.4 vs 16.6 ms)
  float4 aR[PD][NA];
  uint4 wR[PD][NW];
  // (loads are unconditional -- rows and k offsets clamped -- so the compiler's vmcnt counting stays exact: a stage is
  //  written to LDS as soon as ITS loads have landed, with the younger stages still in flight)
  auto stage_load = [&](float4 (&ar)[NA], uint4 (&wr)[NW], int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + i * 256;
      const int row = min(m0 + idx / AC, g.T - 1);  // rows beyond T repeat the last one; their outputs are never stored
      ar[i] = *reinterpret_cast<const float4*>(g.A + (size_t)row * g.K + k0 + (idx % AC) * 4);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = tid + i * 256;
      const int row = min(n0 + idx / WC, g.N - 1);
      wr[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const WT*>(g.wt) + (size_t)row * g.K + k0 + (idx % WC) * (16 / (int)sizeof(WT)));
    }
  };
  auto stage_write = [&](const float4 (&ar)[NA], const uint4 (&wr)[NW]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / AC, c4 = idx % AC;
      if constexpr (F32) {
        float* d = As + row * PITCH + c4 * 4;
        d[0] = ar[i].x; d[1] = ar[i].y; d[2] = ar[i].z; d[3] = ar[i].w;
      } else {
        uint2 pk;
        pk.x = pack_bf16x2(ar[i].x, ar[i].y);
        pk.y = pack_bf16x2(ar[i].z, ar[i].w);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(As) + row * (PITCH * 2) + c4 * 4) = pk;
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / WC, c = idx % WC;
      if constexpr (F32) {
        float* d = Ws + row * PITCH + c * 4;
        d[0] = __uint_as_float(wr[i].x); d[1] = __uint_as_float(wr[i].y); d[2] = __uint_as_float(wr[i].z); d[3] = __uint_as_float(wr[i].w);
      } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(Ws) + row * (PITCH * 2) + c * 8) = wr[i];
      }
    }
  };

  const int nkt = g.K / BK;
#pragma unroll
  for (int s = 0; s < PD; ++s) stage_load(aR[s], wR[s], min(s, nkt - 1) * BK);
  for (int kt0 = 0; kt0 < nkt; kt0 += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      if (kt0 + s >= nkt) break;
      __syncthreads();  // every wave is done with the previous k-tile
      stage_write(aR[s], wR[s]);
      __syncthreads();
      stage_load(aR[s], wR[s], min(kt0 + s + PD, nkt - 1) * BK);
    if constexpr (F32) {
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float a[MI], b[2];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = As[(wm * (BM / 2) + mi * 32 + l31) * PITCH + kk + lh];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Ws[(wn * 64 + j * 32 + l31) * PITCH + kk + lh];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[j], acc[mi][j], 0, 0, 0);
      }
    } else {
      const unsigned short* A16 = reinterpret_cast<const unsigned short*>(As);
      const unsigned short* W16 = reinterpret_cast<const unsigned short*>(Ws);
#pragma unroll
      for (int kk = 0; kk < BK; kk += 16) {
        bf16x8 a[MI], b[2];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(A16 + (wm * (BM / 2) + mi * 32 + l31) * (PITCH * 2) + kk + lh * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(W16 + (wn * 64 + j * 32 + l31) * (PITCH * 2) + kk + lh * 8);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[j], acc[mi][j], 0, 0, 0);
      }
    }
    }
  }
  // ---- epilogue (C layout: col = lane&31 -> n, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> m)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
      if (n >= g.N) continue;
      const float bias = g.bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.T) continue;
        const float v = acc[mi][j][r] + bias;
        if constexpr (EPI == RE_RESID) {
          float* o = g.out + (size_t)m * g.N + n;
          *o = *o + v;
        } else if constexpr (EPI == RE_GELU) {
          g.out[(size_t)m * g.N + n] = gelu_new_f(v);
        } else {
          if (n < g.D) {
            g.out[(size_t)m * g.D + n] = v;
          } else {
            const int which = n / g.D;
            const int c = n - which * g.D;
            const int hh = c / HD, d = c % HD;
            KVT* cache = reinterpret_cast<KVT*>(which == 1 ? g.kcache : g.vcache);
            store_kv(cache + ((size_t)hh * g.smax + g.pos0 + m) * HD + d, v);
          }
        }
      }
    }
}

// ---- bf16 GEMM of the rows path: C[T][N] = A16[T][K] . Wt[N][K]^T (+ epilogue), tile 128 x 128 x 64, 4 waves (2 x 2),
// wave tile 64 x 64 on v_mfma_f32_32x32x16_bf16.  Both operands are bf16 in global memory (the LayerNorm / attention /
// gelu producers write bf16 rows), so a k-tile is 32 one-KiB pieces copied global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, no ds_write pass) into two LDS buffers: the copy of k-tile i+1 is in flight under the
// MFMAs of k-tile i (raw s_barrier + counted vmcnt: a __syncthreads would drain it).  (A ring of 4 buffers measured the
// same: the copies are not what bounds the loop.)  The LDS image is row-linear (what
// the DMA writes); the 16-byte granule g of row r sits at position g ^ ((r >> 1) & 7) -- swizzled on the SOURCE address
// and on the fragment read.  A ds_read_b128 is served in 4 groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32)
// over 64 banks = 16 granules: with 128-byte rows the granule slot is 8 (r & 1) + position, and (r >> 1) & 7 takes 8
// different values over each group's even rows and over its odd rows -> conflict-free (g ^ (r & 7) was 2-way: PMC).
struct GemmArgs16 {
  const unsigned short* A;  // [T][K] bf16
  const unsigned short* wt; // [N][K] bf16
  const float* bias;        // [N]
  void* out;                // RE_QKV: q fp32 [T][D]; RE_RESID: x fp32 [T][N] (+=); RE_GELU: ff bf16 [T][N]
  void* kcache;             // RE_QKV: slot+layer base [H][smax][64]
  void* vcache;
  int T, N, K, pos0, smax, D;
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int G16_NBUF = 2;                       // LDS buffers: NBUF - 1 k-tiles of copies in flight under the MFMAs
constexpr int G16_TILE = 128 * 64 * 2;            // 16 KiB per operand tile
constexpr int G16_SMEM = G16_NBUF * 2 * G16_TILE; // 64 KiB: two workgroups per CU

template <int EPI, typename KVT>
__global__ __launch_bounds__(256) void gemm_rows_bf16_kernel(GemmArgs16 g) {
  constexpr int BM = 128, BN = 128, BK = 64, TILE = G16_TILE, NBUF = G16_NBUF;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
  unsigned char (*smem)[2][TILE] = reinterpret_cast<unsigned char (*)[2][TILE]>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.f;

  // this wave's 4 + 4 DMA pieces of a k-tile: piece c covers rows 8c .. 8c+7, lane -> (row, swizzled granule)
  const unsigned short* asrc[4];
  const unsigned short* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int gran = (lane & 7) ^ ((row >> 1) & 7);
    asrc[i] = g.A + (size_t)min(m0 + row, g.T - 1) * g.K + gran * 8;  // rows beyond T repeat the last one; never stored
    wsrc[i] = g.wt + (size_t)min(n0 + row, g.N - 1) * g.K + gran * 8;
  }
  auto issue = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(asrc[i] + kt * BK, &smem[buf][0][(wave * 4 + i) * 1024]);
      glds16(wsrc[i] + kt * BK, &smem[buf][1][(wave * 4 + i) * 1024]);
    }
  };
  const int nkt = g.K / BK;
#pragma unroll
  for (int p = 0; p < NBUF - 1; ++p) issue(min(p, nkt - 1), p);
  for (int kt = 0; kt < nkt; ++kt) {
    // (past the end the last tile is re-copied into an idle buffer: keeps the outstanding count fixed)
    issue(min(kt + NBUF - 1, nkt - 1), (kt + NBUF - 1) % NBUF);
    static_assert(NBUF == 2, "the immediate below is 8 * (NBUF - 1)");
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // everything but the youngest k-tile (8 pieces) has landed
    __builtin_amdgcn_s_barrier();
    const unsigned char* At = smem[kt % NBUF][0];
    const unsigned char* Wt = smem[kt % NBUF][1];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int row = wm * 64 + mi * 32 + l31;
        a[mi] = *reinterpret_cast<const bf16x8*>(At + row * 128 + (((kk * 2 + lh) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wn * 64 + j * 32 + l31;
        b[j] = *reinterpret_cast<const bf16x8*>(Wt + row * 128 + (((kk * 2 + lh) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[j], acc[mi][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();  // every wave is done with this buffer before the next iteration's copy lands in it
  }
  // ---- epilogue (C layout: col = lane&31 -> n, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> m).  Loads first, all of them,
  // unconditionally (indices clamped): a residual load under the row-bound branch was waited for on the spot, 64 dependent
  // round trips per lane (~25 us per launch whatever the GEMM size).  Only the stores are predicated.
  float bias[2];
  int ncol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    ncol[j] = n0 + wn * 64 + j * 32 + l31;
    bias[j] = g.bias[min(ncol[j], g.N - 1)];
  }
  if constexpr (EPI == RE_RESID) {
    float* xo = reinterpret_cast<float*>(g.out);
    float old[2][2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = min(m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, g.T - 1);
          old[mi][j][r] = xo[(size_t)m * g.N + min(ncol[j], g.N - 1)];
        }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < g.T && ncol[j] < g.N) xo[(size_t)m * g.N + ncol[j]] = old[mi][j][r] + (acc[mi][j][r] + bias[j]);
        }
  } else {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = ncol[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m >= g.T || n >= g.N) continue;
          const float v = acc[mi][j][r] + bias[j];
          if constexpr (EPI == RE_GELU) {
            // gelu_new(v) = v * sigmoid(2u), u = sqrt(2/pi) (v + 0.044715 v^3): one exp instead of tanhf (the result is rounded to bf16)
            const float u2 = 1.5957691216057308f * (v + 0.044715f * v * v * v);
            store_kv(reinterpret_cast<bf16*>(g.out) + (size_t)m * g.N + n, v / (1.0f + __expf(-u2)));
          } else {
            if (n < g.D) {
              reinterpret_cast<float*>(g.out)[(size_t)m * g.D + n] = v;
            } else {
              const int which = n / g.D;
              const int c = n - which * g.D;
              const int hh = c / HD, d = c % HD;
              KVT* cache = reinterpret_cast<KVT*>(which == 1 ? g.kcache : g.vcache);
              store_kv(cache + ((size_t)hh * g.smax + g.pos0 + m) * HD + d, v);
            }
          }
        }
      }
  }
}

// ---- causal attention over rows: grid (H, T); row t attends keys [valid_from, pos0 + t]
struct AttnRowsArgs {
  const float* q;      // [T][D]
  const void* kcache;  // slot+layer base [H][smax][64]
  const void* vcache;
  void* out;           // [T][D], fp32 or bf16 (OT)
  int T, D, smax, pos0, valid_from;
};

template <typename KVT, typename OT = float>
__global__ __launch_bounds__(256) void attn_rows_kernel(AttnRowsArgs a) {
  constexpr int NW = 4;
  using LY = KVLayout<KVT>;
  __shared__ float sm[NW][LY::LPP][2 + LY::DPL];
  const int hh = blockIdx.x, row = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dp = lane % LY::LPP;
  float qv[LY::DPL];
  load_q_slice<KVT>(a.q + (size_t)row * a.D + hh * HD, dp, qv);
  const KVT* kb = reinterpret_cast<const KVT*>(a.kcache) + (size_t)hh * a.smax * HD + dp * LY::DPL;
  const KVT* vb = reinterpret_cast<const KVT*>(a.vcache) + (size_t)hh * a.smax * HD + dp * LY::DPL;
  SoftAcc<LY::DPL> st;
  st.init();
  const int lo0 = a.valid_from, hi0 = a.pos0 + row + 1;
  attn_sweep<KVT, NW, 8>(st, kb, vb, qv, a.smax, wave, lane, [&](int& lo, int& hi) {
    lo = lo0;
    hi = hi0;
  });
  const float o = attn_merge<KVT, NW>(st, sm, wave, lane);
  if (threadIdx.x < 64) store_kv(reinterpret_cast<OT*>(a.out) + (size_t)row * a.D + hh * HD + threadIdx.x, o);
}

// ---- the same attention as a causal flash kernel on the fp32 matrix cores (the latent pass: 1237 rows x 20 heads; the
// row-at-a-time kernel above re-streams every row's whole key prefix from L2: 94 GB per pass, 6.5 of its 10.4 ms).
// Structure of csrc/attn_full.hip (S^T = K Q^T leaves, in accumulator register r of lane l, the score of query l&31 against
// key (r&3)+8(r>>2)+4(l>>5), which is exactly the B operand of O^T += V^T P^T; one query column per lane, online softmax
// in per-lane scalars, scores in the log2 domain) plus: K/V come from the cache in its own type and are widened to fp32
// while staged; key k is visible to query row t iff valid_from <= k <= pos0 + t; a wave skips the tiles that lie
// entirely behind its last query's limit (it still keeps the workgroup's barriers).  One workgroup = 4 waves = 128 rows.
constexpr int FR_KT = 64, FR_KP = HD + 4;

template <typename KVT>
__device__ __forceinline__ void load_row16(const KVT* p, float (&v)[16]);
template <>
__device__ __forceinline__ void load_row16<float>(const float* p, float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 t = reinterpret_cast<const float4*>(p)[i];
    v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
  }
}
template <>
__device__ __forceinline__ void load_row16<bf16>(const bf16* p, float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint4 r = reinterpret_cast<const uint4*>(p)[i];
    v[8 * i] = lo_bf16(r.x); v[8 * i + 1] = hi_bf16(r.x); v[8 * i + 2] = lo_bf16(r.y); v[8 * i + 3] = hi_bf16(r.y);
    v[8 * i + 4] = lo_bf16(r.z); v[8 * i + 5] = hi_bf16(r.z); v[8 * i + 6] = lo_bf16(r.w); v[8 * i + 7] = hi_bf16(r.w);
  }
}

template <typename KVT, typename OT>
__global__ __launch_bounds__(256) void attn_rows_flash_kernel(AttnRowsArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[FR_KT * FR_KP];
  __shared__ __attribute__((aligned(16))) float Vs[FR_KT * HD];
  const int hh = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int qw0 = blockIdx.y * 128 + wave * 32;  // first row of this wave
  const KVT* kb = reinterpret_cast<const KVT*>(a.kcache) + (size_t)hh * a.smax * HD;
  const KVT* vb = reinterpret_cast<const KVT*>(a.vcache) + (size_t)hh * a.smax * HD;
  const int last_key = a.pos0 + a.T - 1;  // last cache row this pass has written

  float qr[HD / 2];  // Q[row l31][d = 8m + 4 lh + i] for k-step 4m + i, scaled by log2(e) / sqrt(64)
  {
    const int qi = min(qw0 + l31, a.T - 1);
    const float* qp = a.q + (size_t)qi * a.D + hh * HD + 4 * lh;
    const float qs = 0.125f * 1.4426950408889634f;
#pragma unroll
    for (int m = 0; m < HD / 8; ++m) {
      const float4 t = *reinterpret_cast<const float4*>(qp + 8 * m);
      qr[4 * m] = t.x * qs; qr[4 * m + 1] = t.y * qs; qr[4 * m + 2] = t.z * qs; qr[4 * m + 3] = t.w * qs;
    }
  }
  f32x16 ot[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[j][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int lim = a.pos0 + qw0 + l31;             // last key this lane's row may see
  const int wave_lim = a.pos0 + min(qw0 + 31, a.T - 1);  // ... and any row of this wave
  const int wg_lim = a.pos0 + min((int)blockIdx.y * 128 + 127, a.T - 1);
  const int t_first = (a.valid_from / FR_KT) * FR_KT;

  for (int t0 = t_first; t0 <= wg_lim; t0 += FR_KT) {
    __syncthreads();
    {  // stage 64 keys x 64 dims of K and V, widened to fp32: each thread one quarter row of each
      const int key = threadIdx.x >> 2, c16 = (threadIdx.x & 3) * 16;
      // rows past the pass repeat its last one, rows before valid_from (left padding: never written, the cache holds whatever the
      // allocation held -- 0 x NaN in the P V product would poison the row) repeat the first valid one; both are masked below
      const int t = min(max(t0 + key, a.valid_from), last_key);
      float kv[16], vv[16];
      load_row16<KVT>(kb + (size_t)t * HD + c16, kv);
      load_row16<KVT>(vb + (size_t)t * HD + c16, vv);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<float4*>(Ks + key * FR_KP + c16 + 4 * i) = make_float4(kv[4 * i], kv[4 * i + 1], kv[4 * i + 2], kv[4 * i + 3]);
        *reinterpret_cast<float4*>(Vs + key * HD + c16 + 4 * i) = make_float4(vv[4 * i], vv[4 * i + 1], vv[4 * i + 2], vv[4 * i + 3]);
      }
    }
    __syncthreads();
    if (t0 > wave_lim) continue;  // wave-uniform: nothing in this tile is visible to this wave's rows

    f32x16 st[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[j][r] = 0.f;
    const float* kp0 = Ks + l31 * FR_KP + 4 * lh;
    const float* kp1 = kp0 + 32 * FR_KP;
    float4 ka[2][2];
    ka[0][0] = *reinterpret_cast<const float4*>(kp0);
    ka[0][1] = *reinterpret_cast<const float4*>(kp1);
#pragma unroll
    for (int m = 0; m < HD / 8; ++m) {
      if (m + 1 < HD / 8) {
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
    // causal / left-padding mask (only the first and the diagonal tiles have masked keys)
    if (t0 < a.valid_from || t0 + FR_KT - 1 > a.pos0 + qw0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key < a.valid_from || key > lim) st[j][r] = -INFINITY;
        }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, st[j][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    // (a tile may be all -inf for a lane -- behind its limit: alpha = 1, weights 0.  A left-padding row, t < valid_from,
    // sees no key at all: its maximum stays -inf and is replaced by 0 in the exponents, so it ends as a row of zeros,
    // finite like everything else that reaches the cache)
    const float mn = fmaxf(m_run, tmax);
    const float mref = (mn == -INFINITY) ? 0.f : mn;
    const float alpha = __builtin_amdgcn_exp2f(m_run - mref);
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pw = __builtin_amdgcn_exp2f(st[j][r] - mref);
        st[j][r] = pw;
        psum += pw;
      }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = mn;
    if (alpha != 1.0f) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[j][r] *= alpha;
    }
    const float* vp = Vs + (4 * lh) * HD + l31;
    float va[2][2];
    va[0][0] = vp[0];
    va[0][1] = vp[32];
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) {
      const int j = s2 >> 4, r = s2 & 15;
      if (s2 + 1 < 32) {
        const int j1 = (s2 + 1) >> 4, r1 = (s2 + 1) & 15;
        const float* vrow = vp + (j1 * 32 + (r1 & 3) + 8 * (r1 >> 2)) * HD;
        va[(s2 + 1) & 1][0] = vrow[0];
        va[(s2 + 1) & 1][1] = vrow[32];
      }
      const float pv = st[j][r];
      ot[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s2 & 1][0], pv, ot[0], 0, 0, 0);
      ot[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s2 & 1][1], pv, ot[1], 0, 0, 0);
    }
  }
  const int qi = qw0 + l31;
  if (qi < a.T) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    OT* op = reinterpret_cast<OT*>(a.out) + (size_t)qi * a.D + hh * HD;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) store_kv(op + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, ot[j][r] * inv);
  }
}

template <typename KVT, typename OT>
static void launch_attn_rows(ixtts_gpt* h, const AttnRowsArgs& a, hipStream_t st) {
  static const bool legacy = getenv("IXTTS_ROWS_ATTN") && !strcmp(getenv("IXTTS_ROWS_ATTN"), "legacy");  // A/B switch
  if (legacy) hipLaunchKernelGGL((attn_rows_kernel<KVT, OT>), dim3(h->H, a.T), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((attn_rows_flash_kernel<KVT, OT>), dim3(h->H, ceil_div(a.T, 128)), dim3(256), 0, st, a);
}

template <typename WT, typename KVT, int EPI>
static void launch_gemm(const GemmArgs& g, hipStream_t st) {
  // 128-row tiles once there are enough rows to fill the GPU with them (latent pass), 64-row tiles for prompts
  if (g.T >= 512) {
    hipLaunchKernelGGL((gemm_rows_kernel<WT, EPI, KVT, 128>), dim3(ceil_div(g.N, GBN), ceil_div(g.T, 128)), dim3(256), 0, st, g);
  } else {
    hipLaunchKernelGGL((gemm_rows_kernel<WT, EPI, KVT, 64>), dim3(ceil_div(g.N, GBN), ceil_div(g.T, 64)), dim3(256), 0, st, g);
  }
}

template <typename WT, typename KVT, int D>
static int forward_rows_t(ixtts_gpt* h, int slot, int T, int pos0, int valid_from, hipStream_t st) {
  const size_t lstride = (size_t)h->slots * D * h->smax * sizeof(KVT);
  const size_t sstride = (size_t)D * h->smax * sizeof(KVT);
  for (int l = 0; l < h->L; ++l) {
    const LayerOff& o = h->lo[l];
    uint8_t* kc = (uint8_t*)h->kc + l * lstride + slot * sstride;
    uint8_t* vc = (uint8_t*)h->vc + l * lstride + slot * sstride;
    hipLaunchKernelGGL(ln_rows_kernel<D>, dim3(ceil_div(T, 4)), dim3(256), 0, st, h->rx, h->rxn, T);
    GemmArgs g;
    g.A = h->rxn; g.wt = A_PTR(o.wqkv); g.bias = A_F32(o.bqkv); g.out = h->rq; g.kcache = kc; g.vcache = vc;
    g.T = T; g.N = 3 * D; g.K = D; g.pos0 = pos0; g.smax = h->smax; g.D = D;
    launch_gemm<WT, KVT, RE_QKV>(g, st);
    AttnRowsArgs a;
    a.q = h->rq; a.kcache = kc; a.vcache = vc; a.out = h->ratt; a.T = T; a.D = D; a.smax = h->smax; a.pos0 = pos0; a.valid_from = valid_from;
    launch_attn_rows<KVT, float>(h, a, st);
    g.A = h->ratt; g.wt = A_PTR(o.wo); g.bias = A_F32(o.bo); g.out = h->rx; g.N = D; g.K = D;
    launch_gemm<WT, KVT, RE_RESID>(g, st);
    hipLaunchKernelGGL(ln_rows_kernel<D>, dim3(ceil_div(T, 4)), dim3(256), 0, st, h->rx, h->rxn, T);
    g.A = h->rxn; g.wt = A_PTR(o.wfc); g.bias = A_F32(o.bfc); g.out = h->rff; g.N = 4 * D; g.K = D;
    launch_gemm<WT, KVT, RE_GELU>(g, st);
    g.A = h->rff; g.wt = A_PTR(o.wpr); g.bias = A_F32(o.bpr); g.out = h->rx; g.N = D; g.K = 4 * D;
    launch_gemm<WT, KVT, RE_RESID>(g, st);
  }
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

// bf16 weights: the producers write bf16 rows (into the same workspaces) and the GEMMs are the LDS-DMA kernel
template <int D>
static int forward_rows_bf16(ixtts_gpt* h, int slot, int T, int pos0, int valid_from, hipStream_t st) {
  using KVT = bf16;
  const size_t lstride = (size_t)h->slots * D * h->smax * sizeof(KVT);
  const size_t sstride = (size_t)D * h->smax * sizeof(KVT);
  bf16* xn16 = reinterpret_cast<bf16*>(h->rxn);
  bf16* att16 = reinterpret_cast<bf16*>(h->ratt);
  bf16* ff16 = reinterpret_cast<bf16*>(h->rff);
  const dim3 blk(256);
  auto grid = [&](int N) { return dim3(ceil_div(N, 128), ceil_div(T, 128)); };
  static bool attr_done = false;
  if (!attr_done) {
    IX_HIP(hipFuncSetAttribute((const void*)gemm_rows_bf16_kernel<RE_QKV, KVT>, hipFuncAttributeMaxDynamicSharedMemorySize, G16_SMEM));
    IX_HIP(hipFuncSetAttribute((const void*)gemm_rows_bf16_kernel<RE_RESID, KVT>, hipFuncAttributeMaxDynamicSharedMemorySize, G16_SMEM));
    IX_HIP(hipFuncSetAttribute((const void*)gemm_rows_bf16_kernel<RE_GELU, KVT>, hipFuncAttributeMaxDynamicSharedMemorySize, G16_SMEM));
    attr_done = true;
  }
  for (int l = 0; l < h->L; ++l) {
    const LayerOff& o = h->lo[l];
    uint8_t* kc = (uint8_t*)h->kc + l * lstride + slot * sstride;
    uint8_t* vc = (uint8_t*)h->vc + l * lstride + slot * sstride;
    hipLaunchKernelGGL((ln_rows_kernel<D, bf16>), dim3(ceil_div(T, 4)), blk, 0, st, h->rx, xn16, T);
    GemmArgs16 g;
    g.A = reinterpret_cast<const unsigned short*>(xn16); g.wt = reinterpret_cast<const unsigned short*>(A_PTR(o.wqkv)); g.bias = A_F32(o.bqkv);
    g.out = h->rq; g.kcache = kc; g.vcache = vc; g.T = T; g.N = 3 * D; g.K = D; g.pos0 = pos0; g.smax = h->smax; g.D = D;
    hipLaunchKernelGGL((gemm_rows_bf16_kernel<RE_QKV, KVT>), grid(g.N), blk, G16_SMEM, st, g);
    AttnRowsArgs a;
    a.q = h->rq; a.kcache = kc; a.vcache = vc; a.out = att16; a.T = T; a.D = D; a.smax = h->smax; a.pos0 = pos0; a.valid_from = valid_from;
    launch_attn_rows<KVT, bf16>(h, a, st);
    g.A = reinterpret_cast<const unsigned short*>(att16); g.wt = reinterpret_cast<const unsigned short*>(A_PTR(o.wo)); g.bias = A_F32(o.bo);
    g.out = h->rx; g.N = D; g.K = D;
    hipLaunchKernelGGL((gemm_rows_bf16_kernel<RE_RESID, KVT>), grid(g.N), blk, G16_SMEM, st, g);
    hipLaunchKernelGGL((ln_rows_kernel<D, bf16>), dim3(ceil_div(T, 4)), blk, 0, st, h->rx, xn16, T);
    g.A = reinterpret_cast<const unsigned short*>(xn16); g.wt = reinterpret_cast<const unsigned short*>(A_PTR(o.wfc)); g.bias = A_F32(o.bfc);
    g.out = ff16; g.N = 4 * D; g.K = D;
    hipLaunchKernelGGL((gemm_rows_bf16_kernel<RE_GELU, KVT>), grid(g.N), blk, G16_SMEM, st, g);
    g.A = reinterpret_cast<const unsigned short*>(ff16); g.wt = reinterpret_cast<const unsigned short*>(A_PTR(o.wpr)); g.bias = A_F32(o.bpr);
    g.out = h->rx; g.N = D; g.K = 4 * D;
    hipLaunchKernelGGL((gemm_rows_bf16_kernel<RE_RESID, KVT>), grid(g.N), blk, G16_SMEM, st, g);
  }
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

int forward_rows(ixtts_gpt* h, int slot, int T, int pos0, int valid_from, hipStream_t st) {
  if (T <= 0) return IXTTS_OK;
  if (h->cfg.weight_dtype == IXTTS_DTYPE_F32) {
    return h->D == 1280 ? forward_rows_t<float, float, 1280>(h, slot, T, pos0, valid_from, st)
                        : forward_rows_t<float, float, 128>(h, slot, T, pos0, valid_from, st);
  }
  return h->D == 1280 ? forward_rows_bf16<1280>(h, slot, T, pos0, valid_from, st) : forward_rows_bf16<128>(h, slot, T, pos0, valid_from, st);
}

int final_norm_rows(ixtts_gpt* h, const float* x, float* y, int T, hipStream_t st) {
  if (T <= 0) return IXTTS_OK;
  if (h->D == 1280)
    hipLaunchKernelGGL(final_norm_rows_kernel<1280>, dim3(ceil_div(T, 4)), dim3(256), 0, st, x, y, T, (const float*)A_F32(h->lnf_w),
                       (const float*)A_F32(h->lnf_b), (const float*)A_F32(h->fn_w), (const float*)A_F32(h->fn_b));
  else
    hipLaunchKernelGGL(final_norm_rows_kernel<128>, dim3(ceil_div(T, 4)), dim3(256), 0, st, x, y, T, (const float*)A_F32(h->lnf_w),
                       (const float*)A_F32(h->lnf_b), (const float*)A_F32(h->fn_w), (const float*)A_F32(h->fn_b));
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

// rows of [start, codes...] embedded with mel positions 0..: x[r] = mel_emb[tok] + mel_pos[r]
__global__ void embed_mel_rows_kernel(float* x, const float* mel_emb, const float* mel_pos, const int32_t* codes, int start_tok,
                                      int rows, int D) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  const int tok = (r == 0) ? start_tok : codes[r - 1];
  for (int i = threadIdx.x; i < D; i += blockDim.x) x[(size_t)r * D + i] = mel_emb[(size_t)tok * D + i] + mel_pos[(size_t)r * D + i];
}

int embed_mel_rows(ixtts_gpt* h, float* x, const int32_t* codes, int rows, hipStream_t st) {
  if (rows <= 0) return IXTTS_OK;
  hipLaunchKernelGGL(embed_mel_rows_kernel, dim3(rows), dim3(256), 0, st, x, (const float*)A_F32(h->mel_emb), (const float*)A_F32(h->mel_pos), codes,
                     h->cfg.start_mel_token, rows, h->D);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

}  // namespace ixtts
