// gpt_kernels.h -- device kernels of the GPT-2 decode step on gfx950 (rows G1-G8).
//
// One decode step = [sampler+embed] -> 24 x {LN1+QKV GEMV (+KV append) | single-query attention |
// out-proj GEMV (+residual) | LN2+FC GEMV (+gelu_new) | MLP-out GEMV (+residual)} ->
// ln_f + final_norm + head GEMV.  Every kernel is HBM-bandwidth/latency-shaped:
//   * weights are stored transposed ([N][K], K contiguous) and streamed exactly once per step
//     as 16-byte-per-lane coalesced loads straight into VGPRs (a read-once operand gains
//     nothing from an LDS round trip); they are issued FIRST-but-one so they fly across the
//     whole prologue;
//   * each wavefront is independent: it owns ROWS consecutive output rows, keeps the
//     activation slice it needs in registers (the ROWS*K elements of a unit tile the input
//     vector exactly ROWS times, so LayerNorm statistics are one DPP reduction / ROWS) --
//     no LDS, no workgroup barrier in the K=model_dim kernels;
//   * LayerNorm gain/bias are folded into the following matrix at load (W' = W diag(g),
//     b' = b + W beta), so the prologue is only (x - mean) * rstd;
//   * reductions are fixed-order DPP butterflies: results are bit-reproducible run to run.
//
// Reference arithmetic: indextts/gpt/transformers_gpt2.py:480-667 (block), :1164 (ln_f);
// indextts/gpt/model_v2.py:53,156-160,185 (embed, final_norm+mel_head).
#pragma once
#include "common.h"

namespace ixtts {

constexpr int HD = 64;  // head dim (asserted at create)

// Developer timeline (IXTTS_TRACE builds only, see tools/trace_decode.py): wave 0 of every workgroup records
// {kernel id, block, wall clock at entry / after the dot products / at exit} (100 MHz constant clock).
#ifdef IXTTS_TRACE
constexpr int TRACE_SLOTS = 2048, TRACE_WGS = 512;  // launches (modulo) x workgroups per launch
static __device__ unsigned long long* g_trace = nullptr;
// wall clock read that cannot move above the computation of `dep` (and, being volatile, keeps its order)
__device__ __forceinline__ long long clock_after(float dep) {
  long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  return t;
}
struct TraceScope {
  long long t0, ta, tb;
  int kid, seq;
  __device__ __forceinline__ TraceScope(int kernel_id, int launch_seq) : ta(0), tb(0), kid(kernel_id), seq(launch_seq) { t0 = clock_after(0.f); }
  __device__ __forceinline__ void inputs(float dep) { ta = clock_after(dep); }  // activations have arrived
  __device__ __forceinline__ void mid(float dep) { tb = clock_after(dep); }     // weights / keys have arrived and are reduced
  __device__ __forceinline__ void end() {
    const unsigned int blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0 && g_trace && blk < TRACE_WGS) {
      const long long t2 = clock_after(0.f);
      unsigned long long* r = g_trace + ((size_t)(seq % TRACE_SLOTS) * TRACE_WGS + blk) * 8;
      r[0] = ((unsigned long long)(kid + 1) << 32) | (unsigned int)seq;
      r[1] = t0; r[2] = ta ? ta : t0; r[3] = tb ? tb : t2; r[4] = t2;
    }
  }
};
#define IXTTS_TRACE_PARAM , int trace_seq
#define IXTTS_TRACE_SEQ trace_seq
#else
struct TraceScope {
  __device__ __forceinline__ TraceScope(int, int) {}
  __device__ __forceinline__ void inputs(float) {}
  __device__ __forceinline__ void mid(float) {}
  __device__ __forceinline__ void end() {}
};
#define IXTTS_TRACE_PARAM
#define IXTTS_TRACE_SEQ 0
#endif

enum { IN_LN = 0, IN_LN2 = 1, IN_PLAIN = 2, IN_ATTN2 = 4, IN_ATTN4 = 5, IN_LN_PART = 6 };  // IN_ATTNn: merge of n split-S attention partials
// IN_LN_PART: the residual stream is h + bias + the 8 per-XCD partial sums the fused MLP kernel of the previous layer left
// (mlp_fused_kernel); the first workgroup writes the completed h back for the later residual add
constexpr int MLP_XCDS = 8;

// split-S attention partials: per (slot, head, split) [m, l, pad, pad, acc[64]] (acc 16-byte aligned)
constexpr int PART_STRIDE = 4 + 64;
constexpr int NSPLIT_MAX = 8;
enum { EPI_QKV = 0, EPI_RESID = 1, EPI_GELU = 2, EPI_LOGITS = 3 };

struct GemvArgs {
  const void* wt;       // [N][K] weights (float or bf16), K contiguous; LN gain pre-folded
  const float* bias;    // [N] (LN bias pre-folded)
  int N;
  int slot0;            // first sequence slot
  const float* xin;     // [slots][K]: residual stream h (IN_LN*), attention output / ff (IN_PLAIN);
                        // IN_ATTN: split-S partials [slots][H][nsplit][PART_STRIDE]
  int nsplit;           // IN_ATTN
  const float* ln_w;    // IN_LN2 only: explicit gain/bias of the FIRST norm (ln_f)
  const float* ln_b;
  float* out;           // EPI_RESID: h (in place add); EPI_GELU: ff; EPI_LOGITS: logits; EPI_QKV: q
  int out_stride;       // floats between slots in `out`
  void* kcache;         // EPI_QKV: layer base [slots][H][smax][64]
  void* vcache;
  const int* cur_len;   // [slots] KV position of the token being forwarded
  int smax;             // KV capacity per (slot, head)
  int heads;
  float* norm_out;      // optional (IN_LN2): ln_f output BEFORE the folded final_norm gain, unused
  unsigned* aux;        // EPI_QKV: fused-MLP arrival counters to clear (or null)
  float* xout;          // IN_LN_PART: where the completed residual stream is published
  const void* pf_ptr;   // IXTTS_PF builds: the next launch's weights; consumer wave w's bytes start at pf_ptr + w * pf_stride
  int pf_stride, pf_per, pf_nwaves;  // pf_per: KiB fetched ahead per consumer wave (<= IXTTS_PF)
};

// ---- the weight stream of the register GEMVs is loaded NON-TEMPORAL (global_load_dwordx4 ... nt): once-read bytes that then do
// not displace the K/V rows (re-read every step: 74 MB at 600 keys) and the activations from L2 / Infinity Cache.  Measured (r03,
// tools/step_ab.py, 1100 steps from 137 keys, tokens identical): B=1 514.6 -> 488.4 us per step, B=2 592.8 -> 577.1, B=3 667.2 ->
// 661.0.  The wide MFMA GEMVs (gpt_wide.h) keep plain loads: nt cost them 1.6 % (B=6 754.7 -> 766.7, B=16 974.0 -> 989.7).
// IXTTS_NT_W=0 builds the plain-load variant for the A/B.
#ifndef IXTTS_NT_W
#define IXTTS_NT_W 1
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT = (IXTTS_NT_W != 0)>
__device__ __forceinline__ uint4 load_w16(const void* p) {
  if constexpr (NT) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *reinterpret_cast<const uint4*>(p);
  }
}
// IXTTS_PF=n (A/B lever, measured and NOT adopted -- profiles/r03_notes.md): every wave of a GEMV launch also loads up to n KiB of
// the NEXT launch's weights, issued right behind its own weight loads so that they land under its dot products, reduction and
// epilogue; nothing uses the values, the lines stay in L2 / Infinity Cache for the consumer (wave w's first `pf_per` KiB live at
// pf_ptr + w * pf_stride).  B=2: 592.8 -> 626.2 / 624.0 / 616.7 us per step at 1 / 2 / 4 KiB per consumer wave (4 KiB = 40 % of the
// FC matrix fetched ahead): the extra loads cost every launch ~0.25 us, and a consumer that finds 40 % of its weights on-die is
// only ~10 us per step faster than one that finds 10 % -- the chain is not waiting for HBM bytes.
#ifndef IXTTS_PF
#define IXTTS_PF 0
#endif
#if IXTTS_PF > 0
#define IXTTS_PF_PARAM , const void* pf_ptr, int pf_stride, int pf_per, int pf_nwaves
#define IXTTS_PF_ARGS(a) , (a).pf_ptr, (a).pf_stride, (a).pf_per, (a).pf_nwaves
struct PfRegs {
  uint4 r[IXTTS_PF];
};
// unconditional loads, clamped (a load under a branch rejoins with a conservative wait-for-everything, see gemv_reg_kernel)
__device__ __forceinline__ void pf_issue(PfRegs& p, const void* ptr, int stride, int per, int nwaves, int gw, int lane) {
  const char* base = reinterpret_cast<const char*>(ptr) + (size_t)min(gw, nwaves - 1) * stride + lane * 16;
#pragma unroll
  for (int j = 0; j < IXTTS_PF; ++j) p.r[j] = *reinterpret_cast<const uint4*>(base + (size_t)min(j, per - 1) * 1024);
}
__device__ __forceinline__ void pf_retire(const PfRegs& p) {
#pragma unroll
  for (int j = 0; j < IXTTS_PF; ++j) asm volatile("" ::"v"(p.r[j].x), "v"(p.r[j].y), "v"(p.r[j].z), "v"(p.r[j].w));
}
#define IXTTS_PF_ISSUE(gw, lane) PfRegs pf_regs; pf_issue(pf_regs, pf_ptr, pf_stride, pf_per, pf_nwaves, gw, lane); __builtin_amdgcn_sched_barrier(0)
#define IXTTS_PF_RETIRE() pf_retire(pf_regs)
#else
#define IXTTS_PF_PARAM
#define IXTTS_PF_ARGS(a)
#define IXTTS_PF_ISSUE(gw, lane)
#define IXTTS_PF_RETIRE()
#endif

// 16 bytes of weights per lane per load, kept RAW in registers until the dot product.
template <typename WT>
struct WVec;
template <>
struct WVec<float> {
  static constexpr int VEC = 4;
  __device__ static __forceinline__ void unpack(const uint4& r, float (&w)[4]) {
    w[0] = __uint_as_float(r.x); w[1] = __uint_as_float(r.y); w[2] = __uint_as_float(r.z); w[3] = __uint_as_float(r.w);
  }
};
template <>
struct WVec<bf16> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ void unpack(const uint4& v, float (&w)[8]) {
    w[0] = lo_bf16(v.x); w[1] = hi_bf16(v.x); w[2] = lo_bf16(v.y); w[3] = hi_bf16(v.y);
    w[4] = lo_bf16(v.z); w[5] = hi_bf16(v.z); w[6] = lo_bf16(v.w); w[7] = hi_bf16(v.w);
  }
};

__device__ __forceinline__ void store_kv(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_kv(bf16* p, float v) { *p = __float2bfloat16(v); }

__device__ __forceinline__ float gelu_new_f(float x) {
  // 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))   (transformers_gpt2.py:571-585, ACT2FN["gelu_new"])
  const float c = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}

// Shared epilogue: lane (r*B + b) owns output row r of the unit for slot b.
template <int K, int ROWS, int B, int EPI, typename KVT>
__device__ __forceinline__ void gemv_epilogue(const GemvArgs& a, int lane, int unit, const float (&tot)[ROWS][B],
                                              float pre_bias, float pre_res, int pre_pos) {
  if (lane < ROWS * B) {
    const int r = lane / B, b = lane % B;
    float mine = 0.f;
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
      for (int bb = 0; bb < B; ++bb)
        if (rr == r && bb == b) mine = tot[rr][bb];
    const int n = unit * ROWS + r;
    if (n < a.N) {
      const int slot = a.slot0 + b;
      const float v = mine + pre_bias;
      if constexpr (EPI == EPI_RESID) {
        a.out[(size_t)slot * a.out_stride + n] = pre_res + v;
      } else if constexpr (EPI == EPI_GELU) {
        a.out[(size_t)slot * a.out_stride + n] = gelu_new_f(v);
      } else if constexpr (EPI == EPI_LOGITS) {
        a.out[(size_t)slot * a.out_stride + n] = v;
      } else {  // EPI_QKV: q -> buffer, k/v -> cache at position cur_len[slot]
        constexpr int D = K;
        if (n < D) {
          a.out[(size_t)slot * a.out_stride + n] = v;
        } else {
          const int which = n / D;  // 1: k, 2: v
          const int c = n - which * D;
          const int hh = c / HD, d = c % HD;
          KVT* cache = reinterpret_cast<KVT*>(which == 1 ? a.kcache : a.vcache);
          store_kv(cache + (((size_t)slot * a.heads + hh) * a.smax + pre_pos) * HD + d, v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Register-resident GEMV (K = model_dim kernels: QKV, out-proj, FC, head).
//
// Timeline of one workgroup (r01 traces, tools/trace_decode.py): every global load is issued in the first ~0.2 us in
// the order activations -> epilogue operands -> weights (vmcnt retires in order, so each consumer waits only for what
// it needs), pinned with sched_barrier(0): left alone the scheduler sinks the small loads to their use at the end of the
// kernel (one more memory round trip) or batches the activation loads; and no load sits under a branch (a predicated load
// rejoins with a conservative "wait for everything").  The activations land after ~1.4 us, the weights stream in behind
// them, so what remains on the critical path is ALU work at one wave per SIMD (4 cycles per wave64 instruction):
//   * XLDS: the workgroup reads the activation rows once (not ROWS copies per wave through the CU's L1), does the
//     LayerNorm statistics once with two block reductions, and hands the normalised rows to the waves through LDS;
//     IN_ATTN2/4 merge the split-S attention partials in the same staging step;
//   * the dot products run on packed fp32 math (v_pk_fma_f32: two FMAs per lane per instruction).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename WT>
__device__ __forceinline__ void unpack2(const uint4& r, f32x2 (&w)[WVec<WT>::VEC / 2]);
template <>
__device__ __forceinline__ void unpack2<float>(const uint4& r, f32x2 (&w)[2]) {
  w[0] = f32x2{__uint_as_float(r.x), __uint_as_float(r.y)};
  w[1] = f32x2{__uint_as_float(r.z), __uint_as_float(r.w)};
}
template <>
__device__ __forceinline__ void unpack2<bf16>(const uint4& r, f32x2 (&w)[4]) {
  w[0] = f32x2{lo_bf16(r.x), hi_bf16(r.x)};
  w[1] = f32x2{lo_bf16(r.y), hi_bf16(r.y)};
  w[2] = f32x2{lo_bf16(r.z), hi_bf16(r.z)};
  w[3] = f32x2{lo_bf16(r.w), hi_bf16(r.w)};
}

// sums of s[0..B) over the workgroup's WPB waves, in a fixed order (bit-reproducible); `red` is a private [WPB][B] LDS region
template <int B, int WPB>
__device__ __forceinline__ void block_sum(float (&s)[B], float* red, int wave, int lane) {
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const float t = wave_sum63(s[b]);
    if (lane == 63) red[wave * B + b] = t;
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < B; ++b) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) t += red[w * B + b];
    s[b] = t;
  }
}

template <typename WT, int K, int ROWS, int UNITS, int B, int INP, int EPI, typename KVT, int WPB = 4, bool XLDS = false>
__global__ __launch_bounds__(64 * WPB) void gemv_reg_kernel(const void* wt, const float* xin, const float* bias, float* out, int N, int slot0, int out_stride, int smax,
                                                        void* kcache, void* vcache, const int* cur_len, int heads, int nsplit,
                                                        const float* ln_w, const float* ln_b, unsigned* aux, float* xout IXTTS_PF_PARAM IXTTS_TRACE_PARAM) {
  // scalar kernel arguments (not a by-value struct): the first 16 dwords are preloaded into SGPRs at wave launch
  // (-amdgpu-kernarg-preload-count), so the first loads do not wait for a kernarg round trip
  GemvArgs a;
  a.wt = wt; a.xin = xin; a.bias = bias; a.out = out; a.N = N; a.slot0 = slot0; a.out_stride = out_stride; a.smax = smax;
  a.kcache = kcache; a.vcache = vcache; a.cur_len = cur_len; a.heads = heads; a.nsplit = nsplit; a.ln_w = ln_w; a.ln_b = ln_b;
  a.norm_out = nullptr;
  // `aux` = the per-XCD arrival counters of this layer's fused MLP launch (two kernel boundaries ahead), cleared here.  IN_LN_PART: ln_w = the partials [8][slots][K], ln_b = the bias to add, nsplit = slots, xout = the
  // OTHER residual buffer, where the completed stream is published (in place would race with the workgroups still reading).
  if (aux != nullptr && blockIdx.x == 0 && threadIdx.x < MLP_XCDS) aux[threadIdx.x * 32] = 0u;
  TraceScope trace(EPI, IXTTS_TRACE_SEQ);
  constexpr int VEC = WVec<WT>::VEC;
  constexpr int PER = 64 * VEC;       // elements per wave-load
  constexpr int NL = ROWS * K / PER;  // loads per lane per unit
  static_assert(ROWS * K % PER == 0, "unit must be a whole number of wave loads");
  static_assert(K % VEC == 0 && K % 4 == 0, "row length must be a multiple of the vector width");
  static_assert(XLDS || INP == IN_PLAIN, "LayerNorm / split-S merge inputs are staged by the workgroup (XLDS)");
  constexpr int NSP = INP == IN_ATTN2 ? 2 : (INP == IN_ATTN4 ? 4 : 1);
  constexpr int NPASS = (INP == IN_LN || INP == IN_LN_PART) ? 1 : (INP == IN_LN2 ? 2 : 0);

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int unit0 = (blockIdx.x * WPB + wave) * UNITS;  // WPB waves per workgroup: chosen so the grid is one WG per CU
  const int n_units = (a.N + ROWS - 1) / ROWS;
  if (!XLDS && unit0 >= n_units) return;  // wave-uniform; only the XLDS kernels have barriers

  // ---- 1. activations: issue the loads
  constexpr int K4 = K / 4, X4 = B * K4, NT = 64 * WPB, XV = (X4 + NT - 1) / NT;  // XLDS: float4 per thread
  f32x2 xr[B][NL][VEC / 2];
  float4 xs4[XLDS ? XV : 1];             // staged element i of this thread: float4 number threadIdx.x + i*NT of [B][K]
  float4 lw4[NPASS == 2 ? XV : 1], lb4[NPASS == 2 ? XV : 1];
  float4 pbias4[INP == IN_LN_PART ? XV : 1], ppart4[INP == IN_LN_PART ? XV : 1][MLP_XCDS];
  float2 pml[NSP > 1 ? XV : 1][NSP];     // split-S partials: (m, l) and the accumulator slice
  float4 pac[NSP > 1 ? XV : 1][NSP];
  if constexpr (XLDS && NSP == 1) {
    const float* xbase = a.xin + (size_t)a.slot0 * K;  // slots are contiguous rows
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = min((int)threadIdx.x + i * NT, X4 - 1);
      xs4[i] = *reinterpret_cast<const float4*>(xbase + (size_t)idx * 4);
      if constexpr (NPASS == 2) {  // explicit affine of the first norm (ln_f); the last norm's affine is folded into W
        lw4[i] = *reinterpret_cast<const float4*>(a.ln_w + (idx % K4) * 4);
        lb4[i] = *reinterpret_cast<const float4*>(a.ln_b + (idx % K4) * 4);
      }
      if constexpr (INP == IN_LN_PART) {
        const int b = idx / K4, k0 = (idx % K4) * 4;
        pbias4[i] = *reinterpret_cast<const float4*>(a.ln_b + k0);
#pragma unroll
        for (int x = 0; x < MLP_XCDS; ++x)
          ppart4[i][x] = *reinterpret_cast<const float4*>(a.ln_w + ((size_t)x * a.nsplit + a.slot0 + b) * K + k0);
      }
    }
  } else if constexpr (XLDS) {
    constexpr int H = K / HD;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = min((int)threadIdx.x + i * NT, X4 - 1);
      const int b = idx / K4, k0 = (idx % K4) * 4;
      const float* p = a.xin + ((size_t)((a.slot0 + b) * H + k0 / HD) * NSP) * PART_STRIDE;
#pragma unroll
      for (int sp = 0; sp < NSP; ++sp) {
        pml[i][sp] = *reinterpret_cast<const float2*>(p + sp * PART_STRIDE);
        pac[i][sp] = *reinterpret_cast<const float4*>(p + sp * PART_STRIDE + 4 + k0 % HD);
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float* xs = a.xin + (size_t)(a.slot0 + b) * K;
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int k0 = (j * PER + lane * VEC) % K;
#pragma unroll
        for (int v4 = 0; v4 < VEC / 4; ++v4) {
          const float4 t = *reinterpret_cast<const float4*>(xs + k0 + v4 * 4);
          xr[b][j][v4 * 2] = f32x2{t.x, t.y};
          xr[b][j][v4 * 2 + 1] = f32x2{t.z, t.w};
        }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 2. epilogue operands of the rows this lane will write (every lane loads, indices clamped)
  float pre_bias[UNITS], pre_res[UNITS];
  int pre_pos = 0;
  const int elane = min(lane, ROWS * B - 1);
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int n = min((unit0 + u) * ROWS + elane / B, a.N - 1);
    pre_bias[u] = a.bias[n];
    pre_res[u] = 0.f;
    if constexpr (EPI == EPI_RESID) pre_res[u] = a.out[(size_t)(a.slot0 + elane % B) * a.out_stride + n];
  }
  if constexpr (EPI == EPI_QKV) pre_pos = a.cur_len[a.slot0 + elane % B];
  __builtin_amdgcn_sched_barrier(0);
  // ---- 3. weight stream (HBM): unconditional, addresses clamped; products of clamped elements are dropped in step 5
  uint4 wraw[UNITS][NL];
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int unit_c = min(unit0 + u, n_units - 1);
    const WT* base = reinterpret_cast<const WT*>(a.wt) + (size_t)unit_c * ROWS * K;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int e = j * PER + lane * VEC;
      wraw[u][j] = load_w16(base + (e < min(ROWS, a.N - unit_c * ROWS) * K ? e : 0));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  IXTTS_PF_ISSUE(blockIdx.x * WPB + wave, lane);
  // ---- 4. workgroup staging: split-S merge | LayerNorm (gain/bias of the norm feeding the matrix are pre-folded) -> LDS
  if constexpr (XLDS) {
    __shared__ __attribute__((aligned(16))) float xsh[B * K];
    __shared__ float red[(NPASS > 0 ? 2 * NPASS : 1) * WPB * B];
    if constexpr (NSP > 1) {
      // flash-decode merge: x = sum_s acc_s e^{m_s - M} / sum_s l_s e^{m_s - M}
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        float M = -INFINITY;
#pragma unroll
        for (int sp = 0; sp < NSP; ++sp) M = fmaxf(M, pml[i][sp].x);
        float L = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int sp = 0; sp < NSP; ++sp) {
          const float w = (pml[i][sp].x > -INFINITY) ? expf(pml[i][sp].x - M) : 0.f;  // an empty split has m = -inf, l = 0
          L = fmaf(pml[i][sp].y, w, L);
          o.x = fmaf(pac[i][sp].x, w, o.x); o.y = fmaf(pac[i][sp].y, w, o.y);
          o.z = fmaf(pac[i][sp].z, w, o.z); o.w = fmaf(pac[i][sp].w, w, o.w);
        }
        const float inv = 1.0f / L;
        xs4[i] = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
      }
    }
    if constexpr (INP == IN_LN_PART) {
      // complete the residual stream: h += bias + sum over the XCDs' partials (fixed order), and publish it once
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        float4 t = xs4[i];
        t.x += pbias4[i].x; t.y += pbias4[i].y; t.z += pbias4[i].z; t.w += pbias4[i].w;
#pragma unroll
        for (int x = 0; x < MLP_XCDS; ++x) {
          t.x += ppart4[i][x].x; t.y += ppart4[i][x].y; t.z += ppart4[i][x].z; t.w += ppart4[i][x].w;
        }
        xs4[i] = t;
        const int idx = threadIdx.x + i * NT;
        if (blockIdx.x == 0 && (X4 % NT == 0 || idx < X4))
          *reinterpret_cast<float4*>(xout + (size_t)a.slot0 * K + (size_t)idx * 4) = t;
      }
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      // element i of this thread belongs to slot (threadIdx.x + i*NT) / K4; elements past the end count for no slot
      float s[B], q[B];
#pragma unroll
      for (int b = 0; b < B; ++b) s[b] = 0.f;
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int idx = threadIdx.x + i * NT;
        const int sl = (X4 % NT == 0 || idx < X4) ? idx / K4 : B;
        const float v = (xs4[i].x + xs4[i].y) + (xs4[i].z + xs4[i].w);
#pragma unroll
        for (int b = 0; b < B; ++b) s[b] += (sl == b) ? v : 0.f;
      }
      block_sum<B, WPB>(s, red + (2 * pass) * WPB * B, wave, lane);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        s[b] *= (1.0f / K);  // mean
        q[b] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int idx = threadIdx.x + i * NT;
        const int sl = (X4 % NT == 0 || idx < X4) ? idx / K4 : B;
        float mean = 0.f;
#pragma unroll
        for (int b = 0; b < B; ++b) mean = (sl == b) ? s[b] : mean;
        xs4[i].x -= mean; xs4[i].y -= mean; xs4[i].z -= mean; xs4[i].w -= mean;
        const float v = fmaf(xs4[i].x, xs4[i].x, xs4[i].y * xs4[i].y) + fmaf(xs4[i].z, xs4[i].z, xs4[i].w * xs4[i].w);
#pragma unroll
        for (int b = 0; b < B; ++b) q[b] += (sl == b) ? v : 0.f;
      }
      block_sum<B, WPB>(q, red + (2 * pass + 1) * WPB * B, wave, lane);
#pragma unroll
      for (int b = 0; b < B; ++b) q[b] = 1.0f / sqrtf(q[b] * (1.0f / K) + 1e-5f);  // rstd
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int idx = threadIdx.x + i * NT;
        const int sl = (X4 % NT == 0 || idx < X4) ? idx / K4 : B;
        float rstd = 0.f;
#pragma unroll
        for (int b = 0; b < B; ++b) rstd = (sl == b) ? q[b] : rstd;
        xs4[i].x *= rstd; xs4[i].y *= rstd; xs4[i].z *= rstd; xs4[i].w *= rstd;
        if (NPASS == 2 && pass == 0) {
          xs4[i].x = fmaf(xs4[i].x, lw4[i].x, lb4[i].x); xs4[i].y = fmaf(xs4[i].y, lw4[i].y, lb4[i].y);
          xs4[i].z = fmaf(xs4[i].z, lw4[i].z, lb4[i].z); xs4[i].w = fmaf(xs4[i].w, lw4[i].w, lb4[i].w);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = threadIdx.x + i * NT;
      if (X4 % NT == 0 || idx < X4) *reinterpret_cast<float4*>(xsh + idx * 4) = xs4[i];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int k0 = (j * PER + lane * VEC) % K;
#pragma unroll
        for (int v4 = 0; v4 < VEC / 4; ++v4) {
          const float4 t = *reinterpret_cast<const float4*>(xsh + b * K + k0 + v4 * 4);
          xr[b][j][v4 * 2] = f32x2{t.x, t.y};
          xr[b][j][v4 * 2 + 1] = f32x2{t.z, t.w};
        }
      }
  }
  trace.inputs(xr[0][0][0].x);
  // ---- 5. dot products (packed fp32), DPP reduction, epilogue
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int unit = unit0 + u;
    if (unit >= n_units) break;
    const int rows_here = min(ROWS, a.N - unit * ROWS);
    float acc[ROWS][B];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int e = j * PER + lane * VEC;
      const int row = (e < rows_here * K) ? e / K : ROWS;  // ROWS: element beyond the matrix (its load was clamped)
      f32x2 w2[VEC / 2];
      unpack2<WT>(wraw[u][j], w2);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        f32x2 d2 = w2[0] * xr[b][j][0];
#pragma unroll
        for (int v = 1; v < VEC / 2; ++v) d2 = __builtin_elementwise_fma(w2[v], xr[b][j][v], d2);
        const float d = d2.x + d2.y;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r][b] += (row == r) ? d : 0.f;
      }
    }
    float tot[ROWS][B];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) tot[r][b] = wave_sum(acc[r][b]);
    if (u == UNITS - 1) trace.mid(tot[0][0]);
    gemv_epilogue<K, ROWS, B, EPI, KVT>(a, lane, unit, tot, pre_bias[u], pre_res[u], pre_pos);
  }
  IXTTS_PF_RETIRE();
  trace.end();
}

// ------------------------------------------------------------------------------------
// LDS-staged GEMV for the long-K matrix (MLP out, K = 4*model_dim): the activation vector
// (20 KB per slot) is shared by the workgroup's 4 waves through LDS; one barrier.
template <typename WT, int K, int ROWS, int UNITS, int B, int EPI, typename KVT, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void gemv_lds_kernel(const void* wt, const float* xin, const float* bias, float* out, int N, int slot0, int out_stride, int smax,
                                                        void* kcache, void* vcache, const int* cur_len, int heads, int nsplit,
                                                        const float* ln_w, const float* ln_b, unsigned* aux, float* xout IXTTS_PF_PARAM IXTTS_TRACE_PARAM) {
  // scalar kernel arguments (not a by-value struct): the first 16 dwords are preloaded into SGPRs at wave launch
  // (-amdgpu-kernarg-preload-count), so the first loads do not wait for a kernarg round trip
  GemvArgs a;
  a.wt = wt; a.xin = xin; a.bias = bias; a.out = out; a.N = N; a.slot0 = slot0; a.out_stride = out_stride; a.smax = smax;
  a.kcache = kcache; a.vcache = vcache; a.cur_len = cur_len; a.heads = heads; a.nsplit = nsplit; a.ln_w = ln_w; a.ln_b = ln_b;
  a.norm_out = nullptr;
  TraceScope trace(4, IXTTS_TRACE_SEQ);
  constexpr int VEC = WVec<WT>::VEC;
  constexpr int PER = 64 * VEC;
  constexpr int NL = ROWS * K / PER;
  static_assert(ROWS * K % PER == 0, "unit must be a whole number of wave loads");
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [B][K]

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int unit0 = (blockIdx.x * WPB + wave) * UNITS;
  const int n_units = (a.N + ROWS - 1) / ROWS;

  // activation vector: global -> registers (issued before the weight stream)
  constexpr int TOT4 = B * K / 4;
  constexpr int NT = 64 * WPB;
  constexpr int XV = (TOT4 + NT - 1) / NT;  // float4 per thread
  const float* xbase = a.xin + (size_t)a.slot0 * K;  // slots are contiguous: [slot0 .. slot0+B) x K
  float4 xv[XV];
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    const int idx = threadIdx.x + i * NT;
    xv[i] = (TOT4 % NT == 0 || idx < TOT4) ? *reinterpret_cast<const float4*>(xbase + (size_t)idx * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __builtin_amdgcn_sched_barrier(0);  // issue order: activations, epilogue operands, weights (see gemv_reg_kernel)
  float pre_bias[UNITS], pre_res[UNITS];
  const int elane = min(lane, ROWS * B - 1);
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int n = min((unit0 + u) * ROWS + elane / B, a.N - 1);
    pre_bias[u] = a.bias[n];
    pre_res[u] = 0.f;
    if constexpr (EPI == EPI_RESID) pre_res[u] = a.out[(size_t)(a.slot0 + elane % B) * a.out_stride + n];
  }
  __builtin_amdgcn_sched_barrier(0);
  uint4 wraw[UNITS][NL];
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int unit = unit0 + u;  // unconditional loads, see gemv_reg_kernel
    const int unit_c = min(unit, n_units - 1);
    const WT* base = reinterpret_cast<const WT*>(a.wt) + (size_t)unit_c * ROWS * K;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int e = j * PER + lane * VEC;
      wraw[u][j] = load_w16(base + (e < min(ROWS, a.N - unit_c * ROWS) * K ? e : 0));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  IXTTS_PF_ISSUE(blockIdx.x * WPB + wave, lane);
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    const int idx = threadIdx.x + i * NT;
    if (TOT4 % NT == 0 || idx < TOT4) *reinterpret_cast<float4*>(xs + idx * 4) = xv[i];
  }
  __syncthreads();
  trace.inputs(xv[0].x);

#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int unit = unit0 + u;
    if (unit >= n_units) break;
    float acc[ROWS][B];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
    const int rows_here = min(ROWS, a.N - unit * ROWS);
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int e = j * PER + lane * VEC;
      const int row0 = e / K;
      const int k0 = e - row0 * K;
      const int row = (e < rows_here * K) ? row0 : ROWS;  // ROWS: element beyond the matrix (its load was clamped)
      float wv[VEC];
      WVec<WT>::unpack(wraw[u][j], wv);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float* x = xs + b * K + k0;
        float d = 0.f;
#pragma unroll
        for (int v4 = 0; v4 < VEC / 4; ++v4) {
          const float4 t = *reinterpret_cast<const float4*>(x + v4 * 4);
          d = fmaf(wv[v4 * 4 + 0], t.x, d);
          d = fmaf(wv[v4 * 4 + 1], t.y, d);
          d = fmaf(wv[v4 * 4 + 2], t.z, d);
          d = fmaf(wv[v4 * 4 + 3], t.w, d);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r][b] += (row == r) ? d : 0.f;
      }
    }
    float tot[ROWS][B];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) tot[r][b] = wave_sum(acc[r][b]);
    if (u == UNITS - 1) trace.mid(tot[0][0]);
    gemv_epilogue<K, ROWS, B, EPI, KVT>(a, lane, unit, tot, pre_bias[u], pre_res[u], 0);
  }
  IXTTS_PF_RETIRE();
  trace.end();
}

// ------------------------------------------------------------------------------------
// Fused MLP of one decode step (bf16 weights, model_dim 1280): LN2 + c_fc + gelu + c_proj in ONE launch, with the hand-off
// between the two matrices made inside each XCD instead of across a kernel boundary.
//
// 256 workgroups x 5 waves, one per CU.  The dispatcher deals workgroups round-robin over the 8 XCDs starting wherever the
// previous dispatch stopped, so workgroup i runs on XCD (i + c) % 8 with c unknown: the kernel reads its XCD from
// HW_REG_XCC_ID and takes j = i / 8 as its index among that XCD's 32 workgroups (8 consecutive workgroups sit on 8
// different XCDs; the engine checks this dealing once, with a probe).  XCD x
// owns the 640 ff rows [640 x, 640 x + 640): its 32 workgroups compute them (20 rows each, exactly the FC GEMV above),
// publish them through the XCD's L2 -- stores, s_waitcnt, one WORKGROUP-scope atomic per workgroup on the XCD's counter;
// the atomics execute in that L2 and the poll is an atomic too, so no line can be stale in an L1 -- and, once all 32
// have arrived (0.7-0.8 us, tools/spike_xcdsync.hip; an agent-scope barrier costs 4.4 us, a kernel boundary ~3 us of
// exit + launch + first-operand latency), each multiplies the XCD's K-slice of c_proj (weights repacked per XCD, loaded into
// registers at entry together with the FC rows) for 40 of the 1280 outputs.  What leaves the kernel are 8 partial sums per
// output, [xcd][slot][1280]; the next layer's QKV kernel adds them to the residual stream while staging (IN_LN_PART).
// Saves one launch per layer and the second activation round trip (FC 5.4 + MLP-out 5.3 us -> 6.x us).
constexpr int MLP_D = 1280, MLP_FF = 5120, MLP_WAVES = 5, MLP_SLICE = MLP_FF / MLP_XCDS, MLP_ROWS_OUT = MLP_D / 32, MLP_KSUB = MLP_SLICE / MLP_WAVES;
static_assert(MLP_SLICE == 640 && MLP_ROWS_OUT == 40 && MLP_KSUB == 128, "partition of the fused MLP");
constexpr unsigned MLP_SPIN_MAX = 1u << 15;  // x ~0.4 us per poll: ~13 ms, three orders of magnitude above a healthy hand-off
constexpr int MLP_CTR_STRIDE = MLP_XCDS * 32;  // uints per layer: one 128-byte line per XCD (the time-out mark follows the last layer's block)

// sum over the 16 lanes of a DPP row; every lane of the row gets it (fixed order)
__device__ __forceinline__ float group_sum16(float v) {
  v += dpp_take<0xB1, 0xf, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xf, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xf, 0xf>(v);  // row_half_mirror
  v += dpp_take<0x140, 0xf, 0xf>(v);  // row_mirror
  return v;
}

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// out[x][n][c] = in[n][640 x + c]   (c_proj weights [1280][5120] -> one contiguous [1280][640] block per XCD)
static __global__ void mlp_repack_pr_kernel(const bf16* __restrict__ in, bf16* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte granule of the output
  constexpr size_t G = (size_t)MLP_XCDS * MLP_D * MLP_SLICE / 8;
  if (i >= G) return;
  const size_t e = i * 8;
  const int c = (int)(e % MLP_SLICE);
  const int n = (int)((e / MLP_SLICE) % MLP_D);
  const int x = (int)(e / ((size_t)MLP_SLICE * MLP_D));
  reinterpret_cast<uint4*>(out)[i] = *reinterpret_cast<const uint4*>(in + (size_t)n * MLP_FF + (size_t)x * MLP_SLICE + c);
}

// map[i] = XCD of workgroup i
static __global__ void mlp_xcc_probe_kernel(unsigned* map) {
  if (threadIdx.x == 0) map[blockIdx.x] = xcc_id();
}

template <int B>
__global__ __launch_bounds__(64 * MLP_WAVES) void mlp_fused_kernel(const bf16* __restrict__ wfc, const float* __restrict__ hin, const float* __restrict__ bfc,
                                                                   float* ff, const bf16* __restrict__ wprx, float* __restrict__ part,
                                                                   unsigned* ctr, unsigned* mark, int slot0, int slots) {
  constexpr int K = MLP_D, ROWS = 2, UNITS = 2, VEC = 8, PER = 64 * VEC, NL = ROWS * K / PER, WPB = MLP_WAVES, NT = 64 * WPB;
  constexpr int K4 = K / 4, X4 = B * K4, XV = (X4 + NT - 1) / NT;
  constexpr int NPR = MLP_ROWS_OUT / 4;  // c_proj loads per lane: 4 rows x 16 lanes per wave-load
  __shared__ __attribute__((aligned(16))) float xsh[B * K];
  __shared__ float red[2 * WPB * B];
  __shared__ float prs[WPB][MLP_ROWS_OUT][B];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = (int)xcc_id() & 7, j = blockIdx.x >> 3;
#ifdef IXTTS_MLP_LOG
  const unsigned long long t_entry = wall_clock64();
  unsigned long long t_ff = 0, t_arrive = 0, t_rel = 0;
  const bool logging = ctr == mark - (size_t)(gridDim.y + 13) * MLP_CTR_STRIDE;  // layer 10 of 24
#endif
  const int unit0 = ((xcd * 32 + j) * WPB + wave) * UNITS;  // ff rows 640 xcd + 20 j + 4 wave ...

  // ---- 1. activations
  float4 xs4[XV];
  const float* xbase = hin + (size_t)slot0 * K;
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    const int idx = min((int)threadIdx.x + i * NT, X4 - 1);
    xs4[i] = *reinterpret_cast<const float4*>(xbase + (size_t)idx * 4);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 2. bias of the ff rows this lane will write
  float pre_bias[UNITS];
  const int elane = min(lane, ROWS * B - 1);
#pragma unroll
  for (int u = 0; u < UNITS; ++u) pre_bias[u] = bfc[(unit0 + u) * ROWS + elane / B];
  __builtin_amdgcn_sched_barrier(0);
  // ---- 3. the c_fc rows of this workgroup (20)
  uint4 wraw[UNITS][NL];
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const bf16* base = wfc + (size_t)(unit0 + u) * ROWS * K;
#pragma unroll
    for (int jj = 0; jj < NL; ++jj) wraw[u][jj] = *reinterpret_cast<const uint4*>(base + jj * PER + lane * VEC);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 4. LayerNorm (gain / bias folded into c_fc) -> LDS, as in gemv_reg_kernel
  {
    float s[B], q[B];
#pragma unroll
    for (int b = 0; b < B; ++b) s[b] = 0.f;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int sl = (X4 % NT == 0 || idx < X4) ? idx / K4 : B;
      const float v = (xs4[i].x + xs4[i].y) + (xs4[i].z + xs4[i].w);
#pragma unroll
      for (int b = 0; b < B; ++b) s[b] += (sl == b) ? v : 0.f;
    }
    block_sum<B, WPB>(s, red, wave, lane);
#pragma unroll
    for (int b = 0; b < B; ++b) {
      s[b] *= (1.0f / K);
      q[b] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int sl = (X4 % NT == 0 || idx < X4) ? idx / K4 : B;
      float mean = 0.f;
#pragma unroll
      for (int b = 0; b < B; ++b) mean = (sl == b) ? s[b] : mean;
      xs4[i].x -= mean; xs4[i].y -= mean; xs4[i].z -= mean; xs4[i].w -= mean;
      const float v = fmaf(xs4[i].x, xs4[i].x, xs4[i].y * xs4[i].y) + fmaf(xs4[i].z, xs4[i].z, xs4[i].w * xs4[i].w);
#pragma unroll
      for (int b = 0; b < B; ++b) q[b] += (sl == b) ? v : 0.f;
    }
    block_sum<B, WPB>(q, red + WPB * B, wave, lane);
#pragma unroll
    for (int b = 0; b < B; ++b) q[b] = 1.0f / sqrtf(q[b] * (1.0f / K) + 1e-5f);
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int sl = (X4 % NT == 0 || idx < X4) ? idx / K4 : B;
      float rstd = 0.f;
#pragma unroll
      for (int b = 0; b < B; ++b) rstd = (sl == b) ? q[b] : rstd;
      xs4[i].x *= rstd; xs4[i].y *= rstd; xs4[i].z *= rstd; xs4[i].w *= rstd;
      if (X4 % NT == 0 || idx < X4) *reinterpret_cast<float4*>(xsh + idx * 4) = xs4[i];
    }
  }
  __syncthreads();
  // the c_proj slice goes out now -- behind the c_fc rows, which are about to be used, and under the dots, the stores and the
  // hand-off (issued at entry it shared the first 3 us of bandwidth with them: the ff rows were ready 1.3 us later)
  uint4 praw[NPR];
  {
    const bf16* base = wprx + ((size_t)xcd * MLP_D + (size_t)j * MLP_ROWS_OUT + (lane >> 4)) * MLP_SLICE + wave * MLP_KSUB + (lane & 15) * VEC;
#pragma unroll
    for (int i = 0; i < NPR; ++i) praw[i] = *reinterpret_cast<const uint4*>(base + (size_t)(4 * i) * MLP_SLICE);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 5. c_fc rows: packed fp32 dots, gelu, ff to memory (this XCD's L2)
  {
    f32x2 xr[B][NL][VEC / 2];
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int jj = 0; jj < NL; ++jj) {
        const int k0 = (jj * PER + lane * VEC) % K;
#pragma unroll
        for (int v4 = 0; v4 < VEC / 4; ++v4) {
          const float4 t = *reinterpret_cast<const float4*>(xsh + b * K + k0 + v4 * 4);
          xr[b][jj][v4 * 2] = f32x2{t.x, t.y};
          xr[b][jj][v4 * 2 + 1] = f32x2{t.z, t.w};
        }
      }
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
      float acc[ROWS][B];
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
#pragma unroll
      for (int jj = 0; jj < NL; ++jj) {
        const int row = (jj * PER + lane * VEC) / K;
        f32x2 w2[VEC / 2];
        unpack2<bf16>(wraw[u][jj], w2);
#pragma unroll
        for (int b = 0; b < B; ++b) {
          f32x2 d2 = w2[0] * xr[b][jj][0];
#pragma unroll
          for (int v = 1; v < VEC / 2; ++v) d2 = __builtin_elementwise_fma(w2[v], xr[b][jj][v], d2);
          const float d = d2.x + d2.y;
#pragma unroll
          for (int r = 0; r < ROWS; ++r) acc[r][b] += (row == r) ? d : 0.f;
        }
      }
      float mine = 0.f;
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const float t = wave_sum(acc[r][b]);
          mine = (lane == r * B + b) ? t : mine;
        }
      if (lane < ROWS * B) {
        const int r = lane / B, b = lane % B;
        ff[(size_t)(slot0 + b) * MLP_FF + (unit0 + u) * ROWS + r] = gelu_new_f(mine + pre_bias[u]);
      }
    }
  }
  // ---- 6. hand-off inside the XCD
#ifdef IXTTS_MLP_LOG
  t_ff = wall_clock64();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's ff stores have reached L2
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* c = ctr + xcd * 32;
    const unsigned ticket = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef IXTTS_MLP_LOG
    t_arrive = wall_clock64();
#endif
    (void)ticket;
    // the poll must be a real read-modify-write (it executes at L2): `fetch_or(c, 0)` is folded into a plain load, which
    // spins on a stale L1 line.  A compare-exchange that can never succeed returns the current value and is not folded.
    unsigned n = 0, seen;
    auto poll = [&]() {
      unsigned expect = 0xffffffffu;
      __hip_atomic_compare_exchange_strong(c, &expect, 0xffffffffu, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return expect;
    };
    while ((seen = poll()) < 32u) {
      if (++n > MLP_SPIN_MAX) {  // never on a healthy run: leave a mark (and what was seen) instead of hanging the queue
        __hip_atomic_fetch_add(mark, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mark + 1 + xcd, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(mark + 16 + xcd, ticket + 1000u * (unsigned)slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
#ifdef IXTTS_MLP_LOG
  if (threadIdx.x == 0) t_rel = wall_clock64();
#endif
  __syncthreads();
  asm volatile("" ::: "memory");
  // ---- 7. the c_proj partials: every lane reads its 8 columns of this XCD's ff slice straight from L2 (first touch by this
  //         CU in this launch: cannot be stale)
  {
    f32x2 xp[B][VEC / 2];
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int v4 = 0; v4 < VEC / 4; ++v4) {
        const float4 t = *reinterpret_cast<const float4*>(ff + (size_t)(slot0 + b) * MLP_FF + xcd * MLP_SLICE + wave * MLP_KSUB + (lane & 15) * VEC + v4 * 4);
        xp[b][v4 * 2] = f32x2{t.x, t.y};
        xp[b][v4 * 2 + 1] = f32x2{t.z, t.w};
      }
#pragma unroll
    for (int i = 0; i < NPR; ++i) {
      f32x2 w2[VEC / 2];
      unpack2<bf16>(praw[i], w2);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        f32x2 d2 = w2[0] * xp[b][0];
#pragma unroll
        for (int v = 1; v < VEC / 2; ++v) d2 = __builtin_elementwise_fma(w2[v], xp[b][v], d2);
        const float d = group_sum16(d2.x + d2.y);
        if ((lane & 15) == 0) prs[wave][4 * i + (lane >> 4)][b] = d;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < MLP_ROWS_OUT * B) {
    const int r = threadIdx.x / B, b = threadIdx.x % B;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) t += prs[w][r][b];
    part[((size_t)xcd * slots + slot0 + b) * MLP_D + j * MLP_ROWS_OUT + r] = t;
  }
#ifdef IXTTS_MLP_LOG
  if (threadIdx.x == 0 && logging) {
    unsigned long long* lg = reinterpret_cast<unsigned long long*>(mark + MLP_CTR_STRIDE) + blockIdx.x * 6;
    lg[0] = xcd; lg[1] = t_entry; lg[2] = t_ff; lg[3] = t_arrive; lg[4] = t_rel; lg[5] = wall_clock64();
  }
#endif
}

// ------------------------------------------------------------------------------------
// Single-query attention: grid (H, B), NW*64 threads.  lane -> (position lane>>2 of a
// 16-position group, 16-dim slice lane&3); each 4-lane group keeps an online-softmax state
// over its positions; states merge with wavefront shuffles, then across waves through LDS.
// Output is the normalised head vector (softmax(q k^T / 8) v), keys in [valid_from, cur_len].
struct AttnArgs {
  const float* q;       // [slots][D]
  const void* kcache;   // layer base: [slots][H][smax][64]
  const void* vcache;
  float* out;           // nsplit == 1: [slots][D] normalised; else partials [slots][H][nsplit][PART_STRIDE]
  const int* cur_len;
  const int* valid_from;
  int slot0, heads, smax, D, nsplit;
};

// ---- attention building blocks -------------------------------------------------------
// Lane layout: every lane owns ONE 16-byte slice of a key/value row, so a wave-instruction
// reads whole rows back to back (bf16: 8 lanes x 8 dims per key, 8 keys = 1 KiB contiguous;
// fp32: 16 lanes x 4 dims, 4 keys).  IT row groups of loads are issued before any is consumed.
template <typename KVT>
struct KVLayout {
  static constexpr int DPL = 16 / sizeof(KVT);  // dims per lane
  static constexpr int LPP = HD / DPL;          // lanes per key position
  static constexpr int PPW = 64 / LPP;          // key positions per wave-instruction
};
__device__ __forceinline__ void kv_unpack(const uint4& r, float (&o)[4]) {
  o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
}
__device__ __forceinline__ void kv_unpack(const uint4& r, float (&o)[8]) {
  o[0] = lo_bf16(r.x); o[1] = hi_bf16(r.x); o[2] = lo_bf16(r.y); o[3] = hi_bf16(r.y);
  o[4] = lo_bf16(r.z); o[5] = hi_bf16(r.z); o[6] = lo_bf16(r.w); o[7] = hi_bf16(r.w);
}

// Online-softmax state of one lane group over the key positions it owns.
template <int DPL>
struct SoftAcc {
  float m, l, acc[DPL];
  __device__ __forceinline__ void init() {
    m = -INFINITY;
    l = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] = 0.f;
  }
};

// Sweep keys [p_lo, p_hi) of one head (positions are absolute cache rows).  Wave `wave` of NW takes
// PPW-row groups round-robin.  The FIRST pass is issued before p_lo / p_hi are known (`bounds` is
// called after the loads are in flight), which takes one dependent global-load latency off the
// critical path; rows outside [p_lo, p_hi) are masked, addresses are clamped to the cache.
template <typename KVT, int NW, int IT, typename BoundsFn>
__device__ __forceinline__ void attn_sweep(SoftAcc<KVLayout<KVT>::DPL>& st, const KVT* kb, const KVT* vb,
                                           const float (&qv)[KVLayout<KVT>::DPL], int smax, int wave, int lane, BoundsFn bounds,
                                           bool speculate = true) {
  using LY = KVLayout<KVT>;
  constexpr int DPL = LY::DPL, LPP = LY::LPP, PPW = LY::PPW;
  const int pg = lane / LPP;
  int p_lo = 0, p_hi = 0;
  bool have_bounds = false;
  int base0 = 0;
  if (!speculate) {  // a split that does not start at row 0: fetch the bounds first and start at its own range
    bounds(p_lo, p_hi);
    have_bounds = true;
    base0 = p_lo;
  }
  for (int base = base0;; base += NW * PPW * IT) {
    if (have_bounds && base >= p_hi) break;
    uint4 kr[IT], vr[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int p = base + (it * NW + wave) * PPW + pg;
      const size_t off = (size_t)min(p, smax - 1) * HD;
      kr[it] = *reinterpret_cast<const uint4*>(kb + off);
      vr[it] = *reinterpret_cast<const uint4*>(vb + off);
    }
    if (!have_bounds) {
      bounds(p_lo, p_hi);
      have_bounds = true;
    }
    float s[IT];
    bool ok[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int p = base + (it * NW + wave) * PPW + pg;
      ok[it] = (p >= p_lo) && (p < p_hi);
      float kv[DPL];
      kv_unpack(kr[it], kv);
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < DPL; ++i) d = fmaf(qv[i], kv[i], d);
#pragma unroll
      for (int o = 1; o < LPP; o <<= 1) d += __shfl_xor(d, o, 64);
      s[it] = ok[it] ? d : -INFINITY;
    }
    float mn = st.m;
#pragma unroll
    for (int it = 0; it < IT; ++it) mn = fmaxf(mn, s[it]);
    if (mn > -INFINITY) {
      const float sc = expf(st.m - mn);  // exp(-inf) = 0 on the first hit
      st.l *= sc;
#pragma unroll
      for (int i = 0; i < DPL; ++i) st.acc[i] *= sc;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const float pw = ok[it] ? expf(s[it] - mn) : 0.f;
        float vv[DPL];
        kv_unpack(vr[it], vv);
        st.l += pw;
        // rows outside the range were read speculatively and may hold anything (NaN/Inf bit patterns): 0 * NaN != 0
#pragma unroll
        for (int i = 0; i < DPL; ++i) st.acc[i] = fmaf(pw, ok[it] ? vv[i] : 0.f, st.acc[i]);
      }
      st.m = mn;
    }
  }
}

// merge the PPW position groups of a wave, then the NW waves through LDS; thread d < 64 gets dim d
template <typename KVT, int NW>
__device__ __forceinline__ float attn_merge(SoftAcc<KVLayout<KVT>::DPL>& st, float (*sm)[KVLayout<KVT>::LPP][2 + KVLayout<KVT>::DPL],
                                            int wave, int lane, float* outM = nullptr, float* outL = nullptr, float* outO = nullptr) {
  using LY = KVLayout<KVT>;
  constexpr int DPL = LY::DPL, LPP = LY::LPP;
  const int pg = lane / LPP, dp = lane % LPP;
#pragma unroll
  for (int o = LPP; o <= 32; o <<= 1) {
    const float m2 = __shfl_xor(st.m, o, 64);
    const float l2 = __shfl_xor(st.l, o, 64);
    const float mn = fmaxf(st.m, m2);
    const float s1 = (st.m > -INFINITY) ? expf(st.m - mn) : 0.f;
    const float s2 = (m2 > -INFINITY) ? expf(m2 - mn) : 0.f;
    st.l = st.l * s1 + l2 * s2;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      const float a2 = __shfl_xor(st.acc[i], o, 64);
      st.acc[i] = st.acc[i] * s1 + a2 * s2;
    }
    st.m = mn;
  }
  if (pg == 0) {
    sm[wave][dp][0] = st.m;
    sm[wave][dp][1] = st.l;
#pragma unroll
    for (int i = 0; i < DPL; ++i) sm[wave][dp][2 + i] = st.acc[i];
  }
  __syncthreads();
  float res = 0.f;
  if (threadIdx.x < 64) {
    const int d = threadIdx.x, dpp = d / DPL, di = d % DPL;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, sm[w][dpp][0]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float mw = sm[w][dpp][0];
      if (mw > -INFINITY) {
        const float sc = expf(mw - M);
        L += sm[w][dpp][1] * sc;
        O += sm[w][dpp][2 + di] * sc;
      }
    }
    res = O / L;
    if (outM) {
      *outM = M;
      *outL = L;
      *outO = O;
    }
  }
  return res;
}

template <typename KVT>
__device__ __forceinline__ void load_q_slice(const float* qp, int dp, float (&qv)[KVLayout<KVT>::DPL]) {
  constexpr int DPL = KVLayout<KVT>::DPL;
#pragma unroll
  for (int i = 0; i < DPL / 4; ++i) {
    const float4 t = reinterpret_cast<const float4*>(qp + dp * DPL)[i];
    qv[4 * i] = t.x * 0.125f; qv[4 * i + 1] = t.y * 0.125f; qv[4 * i + 2] = t.z * 0.125f; qv[4 * i + 3] = t.w * 0.125f;  // 1/sqrt(64)
  }
}

// Any-length fallback (contexts beyond the largest split-S bucket, IXTTS_ATTN=legacy): one workgroup per (head, slot).
template <typename KVT, int NW, int IT>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const float* q, const void* kcache, const void* vcache,
                                                             const int* cur_len, const int* valid_from, int smax, int heads, int slot0,
                                                             int D, float* out, int nsplit_ IXTTS_TRACE_PARAM) {
  using LY = KVLayout<KVT>;
  __shared__ float sm[NW][LY::LPP][2 + LY::DPL];
  TraceScope trace(5, IXTTS_TRACE_SEQ);
  const int hh = blockIdx.x, slot = slot0 + blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dp = lane % LY::LPP;
  // scalar state first (its latency overlaps the first K/V pass), then q, then the K/V stream
  const int cur = cur_len[slot];
  const int vf = valid_from[slot];
  float qv[LY::DPL];
  load_q_slice<KVT>(q + (size_t)slot * D + hh * HD, dp, qv);
  const KVT* kb = reinterpret_cast<const KVT*>(kcache) + ((size_t)slot * heads + hh) * smax * HD + dp * LY::DPL;
  const KVT* vb = reinterpret_cast<const KVT*>(vcache) + ((size_t)slot * heads + hh) * smax * HD + dp * LY::DPL;
  SoftAcc<LY::DPL> st;
  st.init();
  attn_sweep<KVT, NW, IT>(st, kb, vb, qv, smax, wave, lane, [&](int& lo, int& hi) {
    lo = vf;
    hi = cur + 1;
  });
  trace.mid(st.l);
  const float o = attn_merge<KVT, NW>(st, sm, wave, lane);
  if (threadIdx.x < 64) out[(size_t)slot * D + hh * HD + threadIdx.x] = o;
  trace.end();
}

// ------------------------------------------------------------------------------------
// Split-S single-query attention, the default decode path: grid (H, NSP, B), 4 waves.  Workgroup `sp` of a (head, slot)
// owns the 4*PPW-key blocks sp, sp + NSP, sp + 2 NSP, ... (interleaved, so the split is balanced at every context length
// without knowing it), IT0 of them -- the host picks the instantiation whose coverage IT0 * NSP * 4 * PPW holds the
// longest context of the graph it is about to launch (it counts the steps it has issued), so there is no loop:
//   * every K and V load of the workgroup is issued at entry, before cur_len / valid_from have even arrived
//     (addresses clamped to the cache, rows outside [valid_from, cur_len] masked afterwards); one memory round trip;
//   * softmax in two phases over registers (all scores -> workgroup max through LDS -> weights), so no running
//     rescale and no exp in the cross-lane / cross-wave merges: those are plain sums through LDS;
//   * the 8- or 16-lane dot-product reduction runs on DPP (quad_perm / row_half_mirror / row_mirror), not the LDS crossbar.
// Output: the un-normalised partial (m, l, acc[64]) of the split; the out-proj GEMV merges the NSP partials while it
// stages its activations (IN_ATTN2 / IN_ATTN4).  r01 timeline at context 700, B=2: 8.1 us for the 40-workgroup
// online-softmax kernel above (5.5 us of serial passes + 2.1 us of merge tail) -> see DESIGN.md for this one.
template <int LPP>
__device__ __forceinline__ float group_sum(float v) {  // sum over the LPP consecutive lanes of a key; every lane gets it
  v += dpp_take<0xB1, 0xf, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xf, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xf, 0xf>(v);  // row_half_mirror: the other quad of the 8
  if constexpr (LPP == 16) v += dpp_take<0x140, 0xf, 0xf>(v);  // row_mirror: the other 8 of the 16
  return v;
}

template <typename KVT, int IT0, int NSP>
__global__ __launch_bounds__(256) void attn_split_kernel(const float* q, const void* kcache, const void* vcache, const int* cur_len,
                                                         const int* valid_from, int smax, int heads, int slot0, int D, float* part IXTTS_TRACE_PARAM) {
  using LY = KVLayout<KVT>;
  constexpr int DPL = LY::DPL, LPP = LY::LPP, PPW = LY::PPW;
  static_assert(LPP == 8 || LPP == 16, "lane group of a key");
  __shared__ float wmax[4];
  __shared__ __attribute__((aligned(16))) float racc[4][PPW][HD];
  __shared__ float rl[4][PPW];
  TraceScope trace(5, IXTTS_TRACE_SEQ);
  const int hh = blockIdx.x, sp = blockIdx.y, slot = slot0 + blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pg = lane / LPP, dp = lane % LPP;
  // ---- every load of the kernel, in the order it is consumed
  const int cur = cur_len[slot];
  const int vf = valid_from[slot];
  float qv[DPL];
  load_q_slice<KVT>(q + (size_t)slot * D + hh * HD, dp, qv);
  const KVT* kb = reinterpret_cast<const KVT*>(kcache) + ((size_t)slot * heads + hh) * smax * HD + dp * DPL;
  const KVT* vb = reinterpret_cast<const KVT*>(vcache) + ((size_t)slot * heads + hh) * smax * HD + dp * DPL;
  uint4 kr[IT0], vr[IT0];
#pragma unroll
  for (int it = 0; it < IT0; ++it) {
    const int p = ((it * NSP + sp) * 4 + wave) * PPW + pg;
    kr[it] = *reinterpret_cast<const uint4*>(kb + (size_t)min(p, smax - 1) * HD);
  }
#pragma unroll
  for (int it = 0; it < IT0; ++it) {
    const int p = ((it * NSP + sp) * 4 + wave) * PPW + pg;
    vr[it] = *reinterpret_cast<const uint4*>(vb + (size_t)min(p, smax - 1) * HD);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- scores of this lane group's IT0 keys
  float s[IT0];
  float lmax = -INFINITY;
#pragma unroll
  for (int it = 0; it < IT0; ++it) {
    const int p = ((it * NSP + sp) * 4 + wave) * PPW + pg;
    float kv[DPL];
    kv_unpack(kr[it], kv);
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) d = fmaf(qv[i], kv[i], d);
    d = group_sum<LPP>(d);
    s[it] = (p >= vf && p <= cur) ? d : -INFINITY;
    lmax = fmaxf(lmax, s[it]);
  }
#pragma unroll
  for (int o = LPP; o <= 32; o <<= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
  if (lane == 0) wmax[wave] = lmax;
  __syncthreads();
  const float M = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  // ---- weights and the weighted sum of this lane's DPL dims over its keys
  float l = 0.f, acc[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) acc[i] = 0.f;
#pragma unroll
  for (int it = 0; it < IT0; ++it) {
    const bool ok = s[it] > -INFINITY;
    const float pw = ok ? expf(s[it] - M) : 0.f;
    float vv[DPL];
    kv_unpack(vr[it], vv);
    l += pw;
    // rows outside the range were read speculatively and may hold anything (NaN/Inf bit patterns): 0 * NaN != 0
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] = fmaf(pw, ok ? vv[i] : 0.f, acc[i]);
  }
  trace.mid(l);
  // ---- plain sums over the PPW key groups of each wave and the 4 waves, through LDS in a fixed order
#pragma unroll
  for (int i = 0; i < DPL / 4; ++i)
    *reinterpret_cast<float4*>(&racc[wave][pg][dp * DPL + 4 * i]) = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
  if (dp == 0) rl[wave][pg] = l;
  __syncthreads();
  float* pp = part + (((size_t)slot * heads + hh) * NSP + sp) * PART_STRIDE;
  if (threadIdx.x < HD) {
    float o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int g = 0; g < PPW; ++g) o += racc[w][g][threadIdx.x];
    pp[4 + threadIdx.x] = o;
  } else if (threadIdx.x == HD) {
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int g = 0; g < PPW; ++g) L += rl[w][g];
    pp[0] = M;  // an empty split leaves M = -inf, L = 0, acc = 0: the consumer gives it weight 0
    pp[1] = L;
  }
  trace.end();
}

// ------------------------------------------------------------------------------------
// Sampler + embed (rows G8, G1).  One workgroup of 1024 threads per slot.
struct SamplerState {
  float* logits;        // [slots][V]
  uint8_t* seen;        // [slots][V] ids present in the history (fake prefix {1, start} + generated)
  int32_t* tokens;      // [slots][max_new]
  int* gen_count;       // [slots]
  int* cur_len;         // [slots] out: KV position of the token about to be forwarded
  const int* prompt_len;  // [slots] rows in the cache after prefill
  int* finished;        // [slots]
  int* forced;          // [slots] teacher-forced next token or -1
  float* h;             // [slots][D] out: embedding of the chosen token
  const float* mel_emb; // [V][D]
  const float* mel_pos; // [n_pos][D]
  const ixtts_sampler_cfg* cfg;  // device copy
  float* probs_out;     // optional [slots][V]: the processed probability vector (tests), or null
  int V, D, max_new, n_pos, stop, slot0;
};

// float -> uint key with the same ordering (ascending)
__device__ __forceinline__ unsigned int f2key(float x) {
  const unsigned int b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// counter-based uniform in [0,1): splitmix64 of (seed, slot, step)
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned int slot, unsigned int step) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (((unsigned long long)slot << 32) | (unsigned long long)(step + 1));
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// histogram increment with the wavefront's equal bins merged first: log-probabilities crowd into two or three of the 256
// top-byte bins, and 8194 LDS atomics on the same address serialise (r01: 14 us per radix pass); a leader adds the
// population count of each distinct bin instead (1-3 rounds per wave in the crowded passes).
__device__ __forceinline__ void hist_add_aggregated(unsigned int* hist, unsigned int bin, bool active) {
  bool todo = active;
  while (__ballot(todo)) {  // wave-uniform loop
    if (todo) {
      const unsigned int b0 = __builtin_amdgcn_readfirstlane(bin);  // bin of the first lane still to be counted
      const bool same = bin == b0;
      const unsigned long long m = __ballot(same);
      if (same) {
        if ((unsigned int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u)) == 0u)
          atomicAdd(&hist[b0], (unsigned int)__popcll(m));
        todo = false;
      }
    }
  }
}

// One radix-select pass's decision, by wave 0: the highest bin b whose suffix count S(b) = sum_{j >= b} hist[j] reaches `rem`
// (bin 0 if none does); *prefix |= b << shift, *remaining = rem - S(b + 1).  Each lane takes four consecutive bins and the
// lanes' totals go through one DPP scan.  (One thread walking the 256 bins paid one dependent LDS read per bin: ~10 us per
// pass; 256 threads with shuffles and a hand-over between four waves: 0.6-0.8 us.)  Call with the whole workgroup, hist
// complete; synchronised on return.
__device__ __forceinline__ void radix_pick_bin(const unsigned int* hist, unsigned int* prefix, unsigned int* remaining, int shift) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned int rem = *remaining, pre = *prefix;
    const uint4 h = *reinterpret_cast<const uint4*>(hist + 4 * lane);
    const unsigned int mine = h.x + h.y + h.z + h.w;
    const unsigned int incl = wave_scan_u32(mine);
    const unsigned int total = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
    // suffix counts at this lane's bins, highest first
    const unsigned int s3 = total - incl + h.w, s2 = s3 + h.z, s1 = s2 + h.y, s0 = s1 + h.x;
    const unsigned int above = total - incl;  // S(4 * lane + 4)
    int bin = -1;
    unsigned int snext = 0u;
    if (s3 >= rem && above < rem) { bin = 3; snext = above; }
    else if (s2 >= rem && s3 < rem) { bin = 2; snext = s3; }
    else if (s1 >= rem && s2 < rem) { bin = 1; snext = s2; }
    else if (s0 >= rem && s1 < rem) { bin = 0; snext = s1; }
    if (lane == 0 && s0 < rem) { bin = 0; snext = s1; }  // fewer than `rem` keys in all: the lowest bin
    if (bin >= 0) {
      *prefix = pre | ((unsigned int)(4 * lane + bin) << shift);
      *remaining = rem - snext;
    }
  }
  __syncthreads();
}

// ---- block helpers of the 1024-thread sampler kernels
__device__ __forceinline__ float block_max_1024(float v, float* red) {
  v = wave_max_dpp(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = red[i];
  float m = r[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m = fmaxf(m, r[i]);
  return m;
}
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = red[i];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  return s;
}

// TypicalLogitsWarper (indextts/utils/typical_sampling.py:8-30): keep the tokens whose surprise -log p is closest to the
// entropy, in that order, until their probability mass reaches `mass` (ties with the last one kept), at least `min_keep`.
// scores: this thread's PT values of the V-vector (index threadIdx.x + i*1024), -inf = already removed; filtered in place.
// The threshold is the exact key at which the cumulative mass first reaches `mass`: a 4-pass radix select over the key
// bits with per-bin MASS histograms in 2^-48 fixed point (integer atomics: the result does not depend on their order).
struct TypicalScratch {
  float red[16];
  unsigned long long mhist[256], wtot[4], rem;
  unsigned int prefix, nkept, kmin;
};

template <int PT>
__device__ __forceinline__ void typical_filter_1024(float (&scores)[PT], int V, float mass, int min_keep, TypicalScratch& sc) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PT; ++i) mx = fmaxf(mx, scores[i]);
  mx = block_max_1024(mx, sc.red);
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < PT; ++i) se += (scores[i] > -INFINITY) ? expf(scores[i] - mx) : 0.f;
  const float lse = mx + logf(block_sum_1024(se, sc.red));
  float p[PT], nrm[PT];
  float e = 0.f;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    nrm[i] = scores[i] - lse;
    p[i] = (scores[i] > -INFINITY) ? expf(nrm[i]) : 0.f;
    if (p[i] > 0.f) e -= p[i] * nrm[i];  // nansum: 0 * -inf terms are skipped
  }
  const float ent = block_sum_1024(e, sc.red);
  unsigned int key[PT];
  unsigned long long pm[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    key[i] = (scores[i] > -INFINITY) ? __float_as_uint(fabsf(-nrm[i] - ent)) : 0x7f800000u;  // non-negative floats order as their bits
    pm[i] = (unsigned long long)(p[i] * 281474976710656.0f);
  }
  if (t == 0) {
    sc.prefix = 0u;
    sc.rem = (unsigned long long)((double)mass * 281474976710656.0);
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (t < 256) sc.mhist[t] = 0ull;
    __syncthreads();
    const unsigned int prefix = sc.prefix;
    const unsigned long long rem = sc.rem;
    const unsigned int pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
#pragma unroll
    for (int i = 0; i < PT; ++i)
      if (t + i * 1024 < V && pm[i] > 0ull && (key[i] & pmask) == prefix) atomicAdd(&sc.mhist[(key[i] >> shift) & 0xffu], pm[i]);
    __syncthreads();
    // the lowest bin at which the running mass reaches `rem`, by 256 threads (inclusive prefix sums, ascending)
    unsigned long long x = t < 256 ? sc.mhist[t] : 0ull, pre = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int lo = __shfl_up((unsigned int)pre, o, 64), hi = __shfl_up((unsigned int)(pre >> 32), o, 64);
      if (lane >= o) pre += ((unsigned long long)hi << 32) | lo;
    }
    if (t < 256 && lane == 63) sc.wtot[w] = pre;
    __syncthreads();
    if (t < 256) {
      unsigned long long below = 0ull;
      for (int ww = 0; ww < w; ++ww) below += sc.wtot[ww];
      const unsigned long long P = pre + below, Pprev = P - x;
      const bool last_nonempty = (t == 255);  // (a total short of `rem` by rounding: take the top bin)
      if ((Pprev < rem && P >= rem) || (last_nonempty && P < rem)) {
        sc.prefix = prefix | ((unsigned int)t << shift);
        sc.rem = rem - Pprev;
      }
    }
    __syncthreads();
  }
  const unsigned int tau = sc.prefix;
  // min_tokens_to_keep: the `min_keep` smallest keys stay whatever the threshold (only matters when fewer survive)
  if (t == 0) {
    sc.nkept = 0u;
    sc.kmin = 0xffffffffu;
  }
  __syncthreads();
  unsigned int mine = 0u;
#pragma unroll
  for (int i = 0; i < PT; ++i) mine += (t + i * 1024 < V && key[i] <= tau) ? 1u : 0u;
  if (mine) atomicAdd(&sc.nkept, mine);
  __syncthreads();
  unsigned int tau2 = tau;
  if ((int)sc.nkept < min_keep) {  // workgroup-uniform
#pragma unroll
    for (int i = 0; i < PT; ++i)
      if (t + i * 1024 < V && key[i] > tau) atomicMin(&sc.kmin, key[i]);
    __syncthreads();
    tau2 = sc.kmin;
  }
#pragma unroll
  for (int i = 0; i < PT; ++i)
    if (!(key[i] <= tau || key[i] == tau2)) scores[i] = -INFINITY;
  __syncthreads();
}

#ifdef BEAM_DBG
static __device__ unsigned long long g_beam_dbg[64];
#define DBG_TS(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_beam_dbg[i] = wall_clock64(); } while (0)
#define DBG_TSW(i, w) do { if (threadIdx.x == 64 * (w) && blockIdx.x == 0) g_beam_dbg[i] = wall_clock64(); } while (0)
#else
#define DBG_TS(i)
#define DBG_TSW(i, w)
#endif
constexpr int SAMP_MAXK = 128;  // top_k supported on the device
constexpr int SAMP_PT = 9;      // logits per thread (V <= 9216)
constexpr int TOPK_POOL = 512;  // candidates at or above the k-th largest thread maximum

struct TopkScratch {
  alignas(16) unsigned int hist[256];
  unsigned int sel_prefix, sel_remaining;
  float pool_v[TOPK_POOL];
  int pool_i[TOPK_POOL];
  int pool_n;
};

// Key prefix of the k-th largest of the workgroup's keys (N per thread; element i of thread t counts when t + i*1024 < limit):
// MSB radix select over the top 8*PASSES bits (4 passes: the exact key; fewer: its leading bits, the rest zero).  Whole
// 1024-thread workgroup; synchronised on return.
template <int N, int PASSES>
__device__ __forceinline__ unsigned int radix_kth_1024(const unsigned int (&key)[N], int limit, int k, TopkScratch& sc) {
  if (threadIdx.x == 0) {
    sc.sel_prefix = 0u;
    sc.sel_remaining = (unsigned int)k;
  }
  for (int pass = 0; pass < PASSES; ++pass) {
    const int shift = 24 - 8 * pass;
    if (threadIdx.x < 256) sc.hist[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned int prefix = sc.sel_prefix;
    const unsigned int pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const bool in = (int)threadIdx.x + i * 1024 < limit && (key[i] & pmask) == prefix;
      // the leading byte (sign + exponent) crowds into two or three bins: merge the wave's equal bins first; the later bytes
      // spread over all 256 (merging them costs one round per distinct bin: 7 us in the second pass, measured)
      if (pass == 0) hist_add_aggregated(sc.hist, (key[i] >> shift) & 0xffu, in);
      else if (in) atomicAdd(&sc.hist[(key[i] >> shift) & 0xffu], 1u);
    }
    __syncthreads();
    DBG_TS(35 + 2 * pass);
    radix_pick_bin(sc.hist, &sc.sel_prefix, &sc.sel_remaining, shift);
    DBG_TS(36 + 2 * pass);
  }
  return sc.sel_prefix;
}

// TopK for k <= SAMP_MAXK: the scores >= the k-th largest (ties with it kept, at most SAMP_MAXK of them), sorted descending
// (value, then lower id first) into sort_v / sort_i [SAMP_MAXK] (-inf past the survivors); returns how many.  vals: this
// thread's PT scores (index threadIdx.x + i*1024; -inf beyond V and for removed tokens, which never survive).
// The k-th largest of the 1024 per-thread maxima is a lower bound of the k-th largest score (k distinct scores reach it), so
// the scores at or above it -- here: at or above its leading 16 bits -- hold the whole top k plus a handful: two radix
// passes over one key per thread instead of four over PT (36 wave-merged histogram rounds per step were most of the
// sampling kernels' time), then an exact rank sort of that pool, one wave-wide ballot per element.
template <int PT>
__device__ __forceinline__ int topk_sorted_1024(const float (&vals)[PT], int V, int k, TopkScratch& sc, float* sort_v, int* sort_i) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  unsigned int key[PT], tmax[1] = {0u};
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    key[i] = f2key(vals[i]);
    tmax[0] = max(tmax[0], key[i]);
  }
  if (t == 0) sc.pool_n = 0;
  if (t < SAMP_MAXK) sort_v[t] = -INFINITY;
  DBG_TS(30);
  unsigned int thr = radix_kth_1024<1, 2>(tmax, 1024, k, sc);
  DBG_TS(31);
  auto collect = [&](int cap) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const bool hit = t + i * 1024 < V && key[i] >= thr && vals[i] > -INFINITY;
      const unsigned long long m = __ballot(hit);
      if (m) {  // one atomic per wave and register, the positions from the lane count (candidates are a few dozen in all)
        int base = 0;
        if (lane == 0) base = atomicAdd(&sc.pool_n, (int)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
        if (hit && pos < cap) {
          sc.pool_v[pos] = vals[i];
          sc.pool_i[pos] = t + i * 1024;
        }
      }
    }
    __syncthreads();
  };
  collect(TOPK_POOL);
  DBG_TS(32);
  int np = sc.pool_n;
  if (np > TOPK_POOL) {  // (workgroup-uniform) one thread holds many of the large scores: the exact select over every score instead
    __syncthreads();
    if (t == 0) sc.pool_n = 0;
    thr = radix_kth_1024<PT, 4>(key, V, k, sc);  // the exact key of the k-th largest score
    // Scores strictly above it (fewer than k <= SAMP_MAXK) all enter the pool, in any order -- the rank sort below orders them.
    // Scores EQUAL to it (HF keeps every tie; the survivor list holds SAMP_MAXK) fill the rest lowest id first: positions from
    // per-(register, wave) counts scanned in id order, not from the waves' arrival order -- the same survivors on every run.
    bool tie[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const bool in = t + i * 1024 < V && vals[i] > -INFINITY;
      const bool gt = in && key[i] > thr;
      tie[i] = in && key[i] == thr;
      const unsigned long long m = __ballot(gt);
      if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&sc.pool_n, (int)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
        if (gt) {
          sc.pool_v[pos] = vals[i];
          sc.pool_i[pos] = t + i * 1024;
        }
      }
      const unsigned long long mt = __ballot(tie[i]);
      if (lane == 0) sc.hist[i * 16 + w] = (unsigned int)__popcll(mt);  // id = t + i * 1024: ascending in (i, wave, lane)
    }
    __syncthreads();
    if (t == 0) {
      unsigned int run = (unsigned int)sc.pool_n;
      for (int c = 0; c < PT * 16; ++c) {
        const unsigned int n = sc.hist[c];
        sc.hist[c] = run;
        run += n;
      }
      sc.pool_n = (int)min(run, (unsigned int)SAMP_MAXK);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const unsigned long long mt = __ballot(tie[i]);
      const int pos = (int)sc.hist[i * 16 + w] + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mt, 0u));
      if (tie[i] && pos < SAMP_MAXK) {
        sc.pool_v[pos] = vals[i];
        sc.pool_i[pos] = t + i * 1024;
      }
    }
    __syncthreads();
    np = sc.pool_n;
  }
  if (np == 0) return 0;
  // rank of pool element i = how many precede it: the wave's lanes hold the pool, one ballot per element and 64 of them
  const int w_u = __builtin_amdgcn_readfirstlane(w);
  if (np <= 64) {
    const float pv = lane < np ? sc.pool_v[lane] : -INFINITY;
    const int pi = lane < np ? sc.pool_i[lane] : 0x7fffffff;
    for (int i = w_u; i < np; i += 16) {
      const float mv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv), i));
      const int mi = __builtin_amdgcn_readlane(pi, i);
      const int rank = (int)__popcll(__ballot(pv > mv || (pv == mv && pi < mi)));
      if (lane == 0) {
        sort_v[rank] = mv;
        sort_i[rank] = mi;
      }
    }
  } else {
    constexpr int NCH = TOPK_POOL / 64;
    float pv[NCH];
    int pi[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const bool in = lane + c * 64 < np;
      pv[c] = in ? sc.pool_v[lane + c * 64] : -INFINITY;
      pi[c] = in ? sc.pool_i[lane + c * 64] : 0x7fffffff;
    }
    for (int i = w_u; i < np; i += 16) {
      const float mv = sc.pool_v[i];
      const int mi = sc.pool_i[i];
      int rank = 0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) rank += (int)__popcll(__ballot(pv[c] > mv || (pv[c] == mv && pi[c] < mi)));
      if (lane == 0 && rank < SAMP_MAXK) {
        sort_v[rank] = mv;
        sort_i[rank] = mi;
      }
    }
  }
  DBG_TS(33);
  __syncthreads();
  DBG_TS(34);
#ifdef BEAM_DBG
  if (threadIdx.x == 0 && blockIdx.x == 0) g_beam_dbg[40] = np;
#endif
  const float kth = sort_v[min(min(k, np), SAMP_MAXK) - 1];
  return __popcll(__ballot(sort_v[lane] >= kth)) + __popcll(__ballot(sort_v[lane + 64] >= kth));
}

static_assert(SAMP_MAXK == 128, "topk_sorted_1024 counts the survivors as two 64-lane ballots");

// One thread's running sums over an LDS array, in index order, eight loads in flight at a time (a dependent LDS read per
// element was ~100 cycles each: 4-5 us of a sampling kernel).  x holds SAMP_MAXK floats; elements at or past n count as 0.
__device__ __forceinline__ float seq_sum_lds(const float* x, int n) {
  float z = 0.f;
  for (int r0 = 0; r0 < n; r0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[r0 + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) z += (r0 + u < n) ? v[u] : 0.f;
  }
  return z;
}

// TopPLogitsWarper over the n survivors in sort_v (descending), by wave 0 alone (call with threadIdx.x < 64): token r goes
// while the probability mass of r and everything after it, summed from the last one backwards as the reference's ascending
// cumsum does, stays <= 1 - top_p; the first min_keep always stay.  Returns how many stay; ev[SAMP_MAXK] is left holding the
// softmax numerators exp(v - max) (0 past n), *Z_out their sum in index order.  The exponentials and quotients are computed
// two per lane; the two running sums by every lane alike from LDS (wave-local hand-over: LDS serves one wave's accesses in
// order).  On one thread, one dependent LDS read, one division and one branch per survivor, this was 2.7 us.
__device__ __forceinline__ int topp_wave0(const float* sort_v, int n, float top_p, int min_keep, float* ev, float* qv, float* Z_out) {
  const int lane = threadIdx.x;
  const float mx = sort_v[0];
  const float e0 = lane < n ? expf(sort_v[lane] - mx) : 0.f, e1 = lane + 64 < n ? expf(sort_v[lane + 64] - mx) : 0.f;
  ev[lane] = e0;
  ev[lane + 64] = e1;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  const float Z = seq_sum_lds(ev, n);
  *Z_out = Z;
  int keep = n;
  if (top_p < 1.0f && n > 0) {
    qv[lane] = e0 / Z;
    qv[lane + 64] = e1 / Z;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    float tail = 0.f;
    bool go = true;
    for (int r0 = (n - 1) & ~7; r0 >= 0 && go; r0 -= 8) {
      float q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = qv[r0 + u];
#pragma unroll
      for (int u = 7; u >= 0; --u) {
        const int r = r0 + u;
        if (go && r < n && r >= min_keep) {
          tail += q[u];
          if (tail <= 1.0f - top_p) keep = r;
          else go = false;
        }
      }
    }
  }
  return keep;
}

#ifdef IXTTS_ENGINE_TU  // non-template kernels: compiled into gpt_engine.hip only

// Processor chain in `_get_logits_processor` order (generation_utils.py:900-901,1020-1044):
// [suppress] -> RepetitionPenalty -> Temperature -> TopK (ties with the k-th value kept) -> TopP -> softmax ->
// multinomial(1)  |  argmax when do_sample == 0.
__global__ __launch_bounds__(1024) void sampler_kernel(SamplerState s) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  __shared__ int tok_s;
  __shared__ TopkScratch tk;
  __shared__ float sort_v[SAMP_MAXK];
  __shared__ int sort_i[SAMP_MAXK];

  const int slot = s.slot0 + blockIdx.x;
  // every load whose address is known at entry goes out first, unconditionally (clamped): logits, history bitmap, sampler
  // configuration, the slot's counters -- one memory round trip instead of a chain of dependent ones
  const float* lg = s.logits + (size_t)slot * s.V;
  const uint8_t* seen = s.seen + (size_t)slot * s.V;
  float raw[SAMP_PT];
  uint8_t sn[SAMP_PT];
#pragma unroll
  for (int i = 0; i < SAMP_PT; ++i) {
    const int v = min((int)threadIdx.x + i * 1024, s.V - 1);
    raw[i] = lg[v];
    sn[i] = seen[v];
  }
  const ixtts_sampler_cfg cfg = *s.cfg;
  const int st_finished = s.finished[slot], st_forced = s.forced[slot], st_gen = s.gen_count[slot], st_prompt = s.prompt_len[slot];
  __builtin_amdgcn_sched_barrier(0);
  // the positional row of the token about to be chosen is known already (k = gen_count + 1 -> row k + 1)
  const float* pe = s.mel_pos + (size_t)min(st_gen + 2, s.n_pos - 1) * s.D;
  float pe0 = 0.f, pe1 = 0.f;
  if ((int)threadIdx.x < s.D) pe0 = pe[threadIdx.x];
  if ((int)threadIdx.x + 1024 < s.D) pe1 = pe[threadIdx.x + 1024];
  const float theta = cfg.repetition_penalty;
  const bool sampling = cfg.do_sample != 0;
  const float inv_t = (sampling && cfg.temperature > 0.f) ? 1.0f / cfg.temperature : 1.0f;

  float vals[SAMP_PT];
#pragma unroll
  for (int i = 0; i < SAMP_PT; ++i) {
    const int v = threadIdx.x + i * 1024;
    float x = -INFINITY;
    if (v < s.V) {
      x = raw[i];
      if (cfg.suppress_stop && v == s.stop) x = -INFINITY;
      if (sn[i] && theta != 1.0f) x = (x < 0.f) ? x * theta : x / theta;  // RepetitionPenaltyLogitsProcessor
    }
    vals[i] = x;
  }
  if (cfg.typical_mass > 0.f) {  // custom processor of inference_speech(typical_sampling=True): after the penalty, before the warpers
    __shared__ TypicalScratch typ;
    typical_filter_1024<SAMP_PT>(vals, s.V, cfg.typical_mass, 1, typ);
  }
  float best = -INFINITY;
  int besti = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < SAMP_PT; ++i) {
    const int v = threadIdx.x + i * 1024;
    if (v < s.V) {
      if (sampling) vals[i] = vals[i] * inv_t;  // TemperatureLogitsWarper (x / T)
      if (vals[i] > best || (vals[i] == best && v < besti)) {
        best = vals[i];
        besti = v;
      }
    }
  }
  // block argmax with lowest-index tie break (torch.argmax returns the first maximal element): the wave's maximum, then the
  // lowest index among the lanes that hold it (two DPP reductions; twelve ds_bpermute shuffles before)
  {
    const float wmax = wave_max_dpp(best);
    besti = wave_min_i32_dpp(best == wmax ? besti : 0x7fffffff);
    best = wmax;
  }
  if ((threadIdx.x & 63) == 0) {
    bv[threadIdx.x >> 6] = best;
    bi[threadIdx.x >> 6] = besti;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float v16[16];
    int i16[16];
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      v16[w] = bv[w];
      i16[w] = bi[w];
    }
#pragma unroll
    for (int w = 1; w < 16; ++w)
      if (v16[w] > best || (v16[w] == best && i16[w] < besti)) {
        best = v16[w];
        besti = i16[w];
      }
    tok_s = besti;
  }

  const bool general = sampling && (cfg.top_k <= 0 || cfg.top_k > SAMP_MAXK);
  if (general) {
    // ---- top_k = 0 ("off": HF builds no TopK warper), top_k > 128 or >= V: no survivor list -- the filters become thresholds on
    // the whole vocabulary and the draw an inverse CDF over it.  TopK: the k-th largest key by the same 4-pass radix select.
    __shared__ TypicalScratch gs;
    __shared__ int pick_s;
    if (cfg.top_k > 0 && cfg.top_k < s.V) {
      unsigned int key[SAMP_PT];
#pragma unroll
      for (int i = 0; i < SAMP_PT; ++i) key[i] = f2key(vals[i]);
      const unsigned int thr = radix_kth_1024<SAMP_PT, 4>(key, s.V, cfg.top_k, tk);
#pragma unroll
      for (int i = 0; i < SAMP_PT; ++i)
        if (key[i] < thr) vals[i] = -INFINITY;
    }
    // softmax numerators over what is left
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SAMP_PT; ++i) mx = fmaxf(mx, vals[i]);
    mx = block_max_1024(mx, gs.red);
    float e[SAMP_PT], se = 0.f;
#pragma unroll
    for (int i = 0; i < SAMP_PT; ++i) {
      e[i] = (vals[i] > -INFINITY) ? expf(vals[i] - mx) : 0.f;
      se += e[i];
    }
    const float Z = block_sum_1024(se, gs.red);
    if (cfg.top_p < 1.0f) {
      // ---- TopP: in ascending order of score, tokens go while the running probability mass stays <= 1 - top_p; the largest
      // always stays (min_tokens_to_keep = 1).  The first key whose inclusive mass exceeds the budget, by a 4-pass radix
      // select with per-bin MASS histograms in 2^-48 fixed point (integer atomics: independent of their order); tokens
      // that tie with it are kept.
      unsigned int key[SAMP_PT];
      unsigned long long pm[SAMP_PT];
#pragma unroll
      for (int i = 0; i < SAMP_PT; ++i) {
        key[i] = f2key(vals[i]);
        pm[i] = (unsigned long long)((e[i] / Z) * 281474976710656.0f);
      }
      if (threadIdx.x == 0) {
        gs.prefix = 0u;
        gs.rem = (unsigned long long)((double)(1.0f - cfg.top_p) * 281474976710656.0);
      }
      const int t = threadIdx.x, lane = t & 63, w = t >> 6;
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (t < 256) gs.mhist[t] = 0ull;
        __syncthreads();
        const unsigned int prefix = gs.prefix;
        const unsigned long long rem = gs.rem;
        const unsigned int pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
#pragma unroll
        for (int i = 0; i < SAMP_PT; ++i)
          if (t + i * 1024 < s.V && pm[i] > 0ull && (key[i] & pmask) == prefix) atomicAdd(&gs.mhist[(key[i] >> shift) & 0xffu], pm[i]);
        __syncthreads();
        unsigned long long x = t < 256 ? gs.mhist[t] : 0ull, pre = x;  // inclusive prefix sums over the bins, ascending
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned int lo = __shfl_up((unsigned int)pre, o, 64), hi = __shfl_up((unsigned int)(pre >> 32), o, 64);
          if (lane >= o) pre += ((unsigned long long)hi << 32) | lo;
        }
        if (t < 256 && lane == 63) gs.wtot[w] = pre;
        __syncthreads();
        if (t < 256) {
          unsigned long long below = 0ull;
          for (int ww = 0; ww < w; ++ww) below += gs.wtot[ww];
          const unsigned long long P = pre + below, Pprev = P - x;
          // the bin in which the running mass first EXCEEDS the budget (no such bin: the top one, where the maximum sits)
          if ((Pprev <= rem && P > rem) || (t == 255 && P <= rem)) {
            gs.prefix = prefix | ((unsigned int)t << shift);
            gs.rem = rem - Pprev;
          }
        }
        __syncthreads();
      }
      const unsigned int tau = gs.prefix, kmax = f2key(mx);
#pragma unroll
      for (int i = 0; i < SAMP_PT; ++i)
        if (key[i] < tau && key[i] != kmax) e[i] = 0.f;
    }
    // ---- multinomial(1): inverse CDF over the kept tokens in (thread, register) order -- a fixed order, so a fixed stream
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < SAMP_PT; ++i) mine += e[i];
    float inc = mine;  // inclusive scan over the wave, then over the 16 waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float y = __shfl_up(inc, o, 64);
      if ((int)(threadIdx.x & 63) >= o) inc += y;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 63) gs.red[threadIdx.x >> 6] = inc;
    if (threadIdx.x == 0) pick_s = -1;
    __syncthreads();
    float before = 0.f, total = 0.f;
    for (int w = 0; w < 16; ++w) {
      if (w < (int)(threadIdx.x >> 6)) before += gs.red[w];
      total += gs.red[w];
    }
    const float excl = before + inc - mine;
    const float target = uniform01(cfg.seed, (unsigned int)slot, (unsigned int)st_gen) * total;
    if (mine > 0.f && target >= excl && target < excl + mine) {  // at most one thread owns the target
      float c = excl;
      int pk = -1;
#pragma unroll
      for (int i = 0; i < SAMP_PT; ++i) {
        c += e[i];
        if (pk < 0 && e[i] > 0.f && target < c) pk = threadIdx.x + i * 1024;
      }
      if (pk < 0)
#pragma unroll
        for (int i = 0; i < SAMP_PT; ++i)
          if (e[i] > 0.f) pk = threadIdx.x + i * 1024;  // rounding at the thread's upper edge: its last kept token
      pick_s = pk;
    }
    if (s.probs_out) {
      float* po = s.probs_out + (size_t)slot * s.V;
#pragma unroll
      for (int i = 0; i < SAMP_PT; ++i)
        if ((int)threadIdx.x + i * 1024 < s.V) po[threadIdx.x + i * 1024] = e[i] / total;
    }
    __syncthreads();
    if (threadIdx.x == 0 && pick_s >= 0) tok_s = pick_s;  // (target == total by rounding: the argmax already in tok_s)
  } else if (sampling) {
    // ---- TopK: the scores >= the k-th largest (ties kept), sorted descending (value, then lower id first)
    const int k = min(max(cfg.top_k, 1), SAMP_MAXK);
    const int n = topk_sorted_1024<SAMP_PT>(vals, s.V, k, tk, sort_v, sort_i);
    __shared__ float ev[SAMP_MAXK], qv[SAMP_MAXK];
    if (s.probs_out) {
      float* po = s.probs_out + (size_t)slot * s.V;
      for (int v = threadIdx.x; v < s.V; v += 1024) po[v] = 0.f;
      __syncthreads();
    }
    if (threadIdx.x < 64 && n > 0) {
      // ---- TopP over the survivors (keep >= 1), then multinomial(1) by inverse CDF over the kept ones
      float Z;
      const int keep = topp_wave0(sort_v, n, cfg.top_p, 1, ev, qv, &Z);
      const float Zk = seq_sum_lds(ev, keep);
      const float u = uniform01(cfg.seed, (unsigned int)slot, (unsigned int)st_gen) * Zk;
      float c = 0.f;
      int pick = -1;
      for (int r0 = 0; r0 < keep; r0 += 8) {
        float pr[8];
        int id[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          pr[q] = ev[r0 + q];
          id[q] = sort_i[min(r0 + q, SAMP_MAXK - 1)];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (r0 + q < keep) {
            if (s.probs_out && threadIdx.x == 0) s.probs_out[(size_t)slot * s.V + id[q]] = pr[q] / Zk;
            c += pr[q];
            if (pick < 0 && u < c) pick = id[q];
          }
      }
      if (pick < 0) pick = sort_i[keep - 1];
      if (threadIdx.x == 0) tok_s = pick;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tok = tok_s;
    if (st_finished) tok = s.stop;  // finished rows keep emitting pad == stop (generation_utils.py:3255-3256)
    if (st_forced >= 0) {
      tok = st_forced;
      s.forced[slot] = -1;
    }
    const int k = st_gen + 1;  // this is the k-th generated token
    if (k <= s.max_new) s.tokens[(size_t)slot * s.max_new + k - 1] = tok;
    s.seen[(size_t)slot * s.V + tok] = 1;
    if (tok == s.stop) s.finished[slot] = 1;
    s.gen_count[slot] = k;
    s.cur_len[slot] = st_prompt + k - 1;
    tok_s = tok;
  }
  __syncthreads();
  // embed: mel_embedding[tok] + mel_pos_embedding[k + 1]   (model_v2.py:156-160)
  const float* e = s.mel_emb + (size_t)tok_s * s.D;
  float* h = s.h + (size_t)slot * s.D;
  if ((int)threadIdx.x < s.D) h[threadIdx.x] = e[threadIdx.x] + pe0;
  if ((int)threadIdx.x + 1024 < s.D) h[threadIdx.x + 1024] = e[threadIdx.x + 1024] + pe1;
  for (int i = threadIdx.x + 2048; i < s.D; i += 1024) h[i] = e[i] + pe[i];
}

// copy one embedding row into the residual stream of `slot` and set its KV position
__global__ void set_row_kernel(float* h, const float* row, const float* add, int D, int slot, int* cur_len, int pos) {
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < D; i += blockDim.x * gridDim.x)
    h[(size_t)slot * D + i] = row[i] + (add ? add[i] : 0.f);
  if (blockIdx.x == 0 && threadIdx.x == 0) cur_len[slot] = pos;
}

// ---- weight packing (device side)
// [K][N] fp32 (HF Conv1D) -> staging Wt[N][K] fp32
__global__ void pack_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int N) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    int k = k0 + i, n = n0 + threadIdx.x;
    tile[i][threadIdx.x] = (k < K && n < N) ? src[(size_t)k * N + n] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    int n = n0 + i, k = k0 + threadIdx.x;
    if (n < N && k < K) dst[(size_t)n * K + k] = tile[threadIdx.x][i];
  }
}

#endif  // IXTTS_ENGINE_TU

// finalize: dst[n][k] = WT(src[n][k] * g[k]) ; bias[n] += sum_k src[n][k] * beta[k]   (g/beta may be null)
// one wave per output row.
template <typename WT>
__global__ __launch_bounds__(256) void fold_convert_kernel(const float* __restrict__ src, const float* __restrict__ g,
                                                            const float* __restrict__ beta, float* __restrict__ bias,
                                                            WT* __restrict__ dst, int N, int K) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const float* s = src + (size_t)n * K;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = s[k];
    const float wf = g ? w * g[k] : w;
    if (beta) acc = fmaf(w, beta[k], acc);
    if constexpr (sizeof(WT) == 4) dst[(size_t)n * K + k] = wf;
    else dst[(size_t)n * K + k] = __float2bfloat16(wf);
  }
  if (beta) {
    acc = wave_sum(acc);
    if (lane == 0) bias[n] += acc;
  }
}

}  // namespace ixtts
