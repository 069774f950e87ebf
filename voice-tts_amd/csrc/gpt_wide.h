// gpt_wide.h -- the decode-step GEMVs for WIDE batches (5..16 sequences stepped together) on the bf16 matrix cores.
//
// The register GEMVs of gpt_kernels.h keep one copy of the activations per sequence in every lane, which stops at 4 sequences.
// Beyond that the product is a skinny GEMM  D[row][slot] = W[row][:] . X[slot][:]  with the batch on the N side of
// v_mfma_f32_16x16x32_bf16 (16 columns = 16 sequence slots, whatever the number in use: the columns are independent, so a
// sequence's arithmetic never depends on its company).  The weights stay in the [N][K] bf16 layout of the arena and are still
// streamed exactly once, 16 bytes per lane, every load of the kernel issued up front (activations -> epilogue operands ->
// weights, as in gemv_reg_kernel).
//
// One workgroup = 4 waves = RP0 + RP1 consecutive output rows in one or two 16-row MFMA tiles (one workgroup per CU for the production shapes:
// 15 / 5 / 16 + 4 / 5 / 16 + 16 rows for QKV / out-proj / FC / MLP-out / head).  The K dimension is dealt to the waves in 32-element
// k-steps, wave w owning steps w, w+4, w+8, ...: lane (c = lane & 15, g = lane >> 4) of wave w supplies
//   A: W[row0 + 16 t + c][32 s + 8 g .. +8]   (16 bytes straight from the arena; rows beyond the workgroup's range repeat the last one)
//   B: X[slot c][32 s + 8 g .. +8]            (8 activations of sequence c)
// for s = w + 4 i.  The staging threads are laid out the same way -- thread (c, q = tid >> 4) loads the 8-float chunks
// q, q+16, q+32, ... of sequence c, and (q >> 2) is its wave, (q & 3) its g -- so after the LayerNorm statistics (a two-shuffle
// wave sum + a 4-entry LDS sum per sequence, fixed order) every lane already HOLDS its own B fragments: no LDS image of the
// activations, no bank conflicts.  fp32 activations keep their precision on the bf16 matrix cores as a hi + lo pair
// (x = bf16(x) + bf16(x - bf16(x)), two MFMAs per weight fragment: the products are exact, the sum is fp32) -- the matrix
// pipe is idle otherwise (two 16x16x32 MFMAs per KiB of weights).  The 4 K-partials of a workgroup are summed through LDS in
// a fixed order, then thread (m, n) finishes row m of slot n: bias, residual add, gelu_new, K/V append, as gemv_epilogue.
// The ff activations (K = 4 D, MLP-out) travel as bf16 (hi only): the FC epilogue writes them rounded once, the MLP-out
// waves load their B fragments straight from that buffer.
//
// Reference arithmetic: indextts/gpt/transformers_gpt2.py:480-667 (block), :1164 (ln_f); model_v2.py:53,185 (final_norm + mel_head).
#pragma once
#include "gpt_kernels.h"

namespace ixtts {

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));

enum { WIN_LN = 0, WIN_LN2 = 1, WIN_PLAIN = 2, WIN_FF = 3 };
constexpr int WIDE_COLS = 16;  // sequence slots a wide launch can carry (the N of the MFMA)

// sum over the 4 lanes that hold one sequence in a wave (lanes c, c+16, c+32, c+48), every one of them gets it; fixed order
__device__ __forceinline__ float seq_sum4(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// per-sequence sum over the workgroup: 4 lanes per wave (seq_sum4), then the 4 waves through `red` [4][16]
__device__ __forceinline__ float seq_block_sum(float v, float* red, int wave, int lane) {
  v = seq_sum4(v);
  if (lane < WIDE_COLS) red[wave * WIDE_COLS + lane] = v;
  __syncthreads();
  const int c = lane & 15;
  return (red[c] + red[WIDE_COLS + c]) + (red[2 * WIDE_COLS + c] + red[3 * WIDE_COLS + c]);
}

// lane i of each 16-lane row takes the dword of lane i + N of that row (lanes shifted in from beyond the row read 0)
template <int N>
__device__ __forceinline__ uint4 row_shl4(const uint4& v) {
  if constexpr (N == 0) return v;
  uint4 r;
  r.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x100 + N, 0xf, 0xf, true);
  r.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x100 + N, 0xf, 0xf, true);
  r.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x100 + N, 0xf, 0xf, true);
  r.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x100 + N, 0xf, 0xf, true);
  return r;
}

// RP0 / RP1: output rows the workgroup owns in tile 0 / 1 (rpw = RP0 + RP1 for a two-tile kernel, RP0 alone otherwise).  A tile
// with few rows would waste most of every 16-row weight load (5 rows x 64 B per wave-instruction instead of 1 KiB), so its
// loads are PACKED: 16 / RP k-steps per instruction, lane (m, g) fetching row m % RP of k-step pack m / RP, and the fragment of
// pack p is moved into place with a DPP row shift (4 v_mov_dpp) when its MFMA is issued.
template <int K, int NT, int INP, int EPI, typename KVT, int NW, int RP0, int RP1>
__global__ __launch_bounds__(64 * NW) void gemv_wide_kernel(const bf16* __restrict__ wt, const void* __restrict__ xin, const float* __restrict__ bias, void* out,
                                                        int N, int B, int slot0, int out_stride, int smax, void* kcache, void* vcache,
                                                        const int* __restrict__ cur_len, int heads, const float* __restrict__ ln_w,
                                                        const float* __restrict__ ln_b) {
  static_assert(K % (32 * NW) == 0, "K is dealt to the waves in 32-element k-steps");
  static_assert(NW == 4 || INP == WIN_FF, "the in-register staging of fp32 activations is laid out for 4 waves");
  constexpr int NI = K / (32 * NW);  // k-steps per wave = 8-float chunks per staging thread
  constexpr int NPASS = INP == WIN_LN ? 1 : (INP == WIN_LN2 ? 2 : 0);
  static_assert(RP0 >= 1 && RP0 <= 16 && RP1 >= 0 && RP1 <= 16 && (NT == 2) == (RP1 > 0), "rows per tile");
  constexpr int KP0 = 16 / RP0, KP1 = RP1 > 0 ? 16 / RP1 : 1;              // k-steps per weight load
  constexpr int NL0 = (NI + KP0 - 1) / KP0, NL1 = (NI + KP1 - 1) / KP1;      // weight loads per lane
  constexpr int NLM = NL0 > NL1 ? NL0 : NL1;
  __shared__ float red[4 * NPASS + 1][4 * WIDE_COLS];
  __shared__ __attribute__((aligned(16))) float part[NW][NT][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * (RP0 + RP1);
  const int slot_c = slot0 + min(c, B - 1);  // columns beyond the batch repeat the last sequence (finite, never stored)

  // ---- 1. activations of this lane: chunk (4 wave + g) + 16 i of sequence c  ==  k-step wave + 4 i, elements 8 g .. 8 g + 7
  float x[INP == WIN_FF ? 1 : NI][8];
  float lw[NPASS == 2 ? NI : 1][8], lb[NPASS == 2 ? NI : 1][8];
  uint4 braw[INP == WIN_FF ? NI : 1];
  if constexpr (INP == WIN_FF) {
    const bf16* xb = reinterpret_cast<const bf16*>(xin) + (size_t)slot_c * K + g * 8;
#pragma unroll
    for (int i = 0; i < NI; ++i) braw[i] = *reinterpret_cast<const uint4*>(xb + (wave + NW * i) * 32);
  } else {
    const float* xb = reinterpret_cast<const float*>(xin) + (size_t)slot_c * K + g * 8;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k0 = (wave + NW * i) * 32;
      const float4 t0 = *reinterpret_cast<const float4*>(xb + k0), t1 = *reinterpret_cast<const float4*>(xb + k0 + 4);
      x[i][0] = t0.x; x[i][1] = t0.y; x[i][2] = t0.z; x[i][3] = t0.w; x[i][4] = t1.x; x[i][5] = t1.y; x[i][6] = t1.z; x[i][7] = t1.w;
      if constexpr (NPASS == 2) {  // explicit affine of the first norm (ln_f); the second norm's affine is folded into W
        const float4 w0 = *reinterpret_cast<const float4*>(ln_w + k0 + g * 8), w1 = *reinterpret_cast<const float4*>(ln_w + k0 + g * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(ln_b + k0 + g * 8), b1 = *reinterpret_cast<const float4*>(ln_b + k0 + g * 8 + 4);
        lw[i][0] = w0.x; lw[i][1] = w0.y; lw[i][2] = w0.z; lw[i][3] = w0.w; lw[i][4] = w1.x; lw[i][5] = w1.y; lw[i][6] = w1.z; lw[i][7] = w1.w;
        lb[i][0] = b0.x; lb[i][1] = b0.y; lb[i][2] = b0.z; lb[i][3] = b0.w; lb[i][4] = b1.x; lb[i][5] = b1.y; lb[i][6] = b1.z; lb[i][7] = b1.w;
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 2. epilogue operands: thread (m = tid >> 4, n = tid & 15) finishes row 16 t + m of sequence n
  const int em = (tid >> 4) & 15, en = tid & 15;  // (threads 256.. of an 8-wave workgroup mirror 0..255 and store nothing)
  const int eslot = slot0 + min(en, B - 1);
  float pre_bias[NT], pre_res[NT];
  int pre_pos = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = min(row0 + (t == 0 ? 0 : RP0) + min(em, (t == 0 ? RP0 : RP1) - 1), N - 1);
    pre_bias[t] = bias[n];
    pre_res[t] = 0.f;
    if constexpr (EPI == EPI_RESID) pre_res[t] = reinterpret_cast<const float*>(out)[(size_t)eslot * out_stride + n];
  }
  if constexpr (EPI == EPI_QKV) pre_pos = cur_len[eslot];
  __builtin_amdgcn_sched_barrier(0);
  // ---- 3. weight stream: load j of tile t = k-steps j KP .. j KP + KP - 1 of the rows of that tile (rows / packs beyond the
  // range repeat a valid address; their products land in accumulator rows nobody reads)
  uint4 a[NT][NLM];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int RP = t == 0 ? RP0 : RP1, KP = t == 0 ? KP0 : KP1, NL = t == 0 ? NL0 : NL1;
    const int pk = min(c / RP, KP - 1);
    const bf16* wrow = wt + (size_t)min(row0 + (t == 0 ? 0 : RP0) + c % RP, N - 1) * K + g * 8;
#pragma unroll
    for (int j = 0; j < NL; ++j) a[t][j] = load_w16<false>(wrow + (wave + NW * min(j * KP + pk, NI - 1)) * 32);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 4. LayerNorm of sequence c over its 16 threads (4 lanes in each of the 4 waves), in registers
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) s += ((x[i][0] + x[i][1]) + (x[i][2] + x[i][3])) + ((x[i][4] + x[i][5]) + (x[i][6] + x[i][7]));
    const float mean = seq_block_sum(s, red[2 * pass], wave, lane) * (1.0f / K);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[i][j] -= mean;
        q = fmaf(x[i][j], x[i][j], q);
      }
    const float rstd = 1.0f / sqrtf(seq_block_sum(q, red[2 * pass + 1], wave, lane) * (1.0f / K) + 1e-5f);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[i][j] *= rstd;
        if (NPASS == 2 && pass == 0) x[i][j] = fmaf(x[i][j], lw[i][j], lb[i][j]);
      }
  }
  // ---- 5. matrix cores: acc[t] (rows 4 g .. 4 g + 3 of tile t, sequence c) += W fragment . (hi + lo) activations
  wf32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = wf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    wbf16x8 bh, bl;
    if constexpr (INP == WIN_FF) {
      bh = __builtin_bit_cast(wbf16x8, braw[i]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bh[j] = (__bf16)x[i][j];
        bl[j] = (__bf16)(x[i][j] - (float)bh[j]);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      constexpr int dummy = 0;
      (void)dummy;
      const int KP = t == 0 ? KP0 : KP1;
      uint4 raw;
      // (i, t are compile-time after unrolling: the switch folds to one shift)
      switch ((i % KP) * (t == 0 ? RP0 : RP1)) {
        case 0: raw = a[t][i / KP]; break;
        case 1: raw = row_shl4<1>(a[t][i / KP]); break;
        case 2: raw = row_shl4<2>(a[t][i / KP]); break;
        case 3: raw = row_shl4<3>(a[t][i / KP]); break;
        case 4: raw = row_shl4<4>(a[t][i / KP]); break;
        case 5: raw = row_shl4<5>(a[t][i / KP]); break;
        case 6: raw = row_shl4<6>(a[t][i / KP]); break;
        case 7: raw = row_shl4<7>(a[t][i / KP]); break;
        case 8: raw = row_shl4<8>(a[t][i / KP]); break;
        case 9: raw = row_shl4<9>(a[t][i / KP]); break;
        case 10: raw = row_shl4<10>(a[t][i / KP]); break;
        case 11: raw = row_shl4<11>(a[t][i / KP]); break;
        case 12: raw = row_shl4<12>(a[t][i / KP]); break;
        case 13: raw = row_shl4<13>(a[t][i / KP]); break;
        case 14: raw = row_shl4<14>(a[t][i / KP]); break;
        default: raw = row_shl4<15>(a[t][i / KP]); break;
      }
      const wbf16x8 af = __builtin_bit_cast(wbf16x8, raw);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bh, acc[t], 0, 0, 0);
      if constexpr (INP != WIN_FF) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bl, acc[t], 0, 0, 0);
    }
  }
  // ---- 6. the 4 K-partials through LDS (fixed order), then the epilogue of (row 16 t + em, sequence en)
#pragma unroll
  for (int t = 0; t < NT; ++t) *reinterpret_cast<float4*>(&part[wave][t][lane][0]) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  __syncthreads();
  const int src_lane = (em >> 2) * 16 + en, src_j = em & 3;  // C/D map: col = lane & 15, row = (lane >> 4) * 4 + j
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = row0 + (t == 0 ? 0 : RP0) + em;  // accumulator row em of tile t
    if (tid < 256 && em < (t == 0 ? RP0 : RP1) && n < N && en < B) {
      float v = (part[0][t][src_lane][src_j] + part[1][t][src_lane][src_j]) + (part[2][t][src_lane][src_j] + part[3][t][src_lane][src_j]);
      if constexpr (NW == 8) v += (part[4][t][src_lane][src_j] + part[5][t][src_lane][src_j]) + (part[6][t][src_lane][src_j] + part[7][t][src_lane][src_j]);
      v += pre_bias[t];
      if constexpr (EPI == EPI_RESID) {
        reinterpret_cast<float*>(out)[(size_t)eslot * out_stride + n] = pre_res[t] + v;
      } else if constexpr (EPI == EPI_GELU) {
        reinterpret_cast<bf16*>(out)[(size_t)eslot * out_stride + n] = __float2bfloat16(gelu_new_f(v));
      } else if constexpr (EPI == EPI_LOGITS) {
        reinterpret_cast<float*>(out)[(size_t)eslot * out_stride + n] = v;
      } else {  // EPI_QKV: q -> buffer, k / v -> cache at position cur_len[slot]  (K == model_dim here)
        if (n < K) {
          reinterpret_cast<float*>(out)[(size_t)eslot * out_stride + n] = v;
        } else {
          const int which = n / K, cc = n - which * K;
          KVT* cache = reinterpret_cast<KVT*>(which == 1 ? kcache : vcache);
          store_kv(cache + (((size_t)eslot * heads + cc / HD) * smax + pre_pos) * HD + cc % HD, v);
        }
      }
    }
  }
}

}  // namespace ixtts
