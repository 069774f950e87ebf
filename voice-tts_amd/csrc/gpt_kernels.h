// gpt_kernels.h -- device kernels of the GPT-2 decode step on gfx950 (rows G1-G8).
//
// One decode step = [sampler+embed] -> 24 x {LN1+QKV GEMV (+KV append) | split-S attention |
// combine+out-proj GEMV (+residual) | LN2+FC GEMV (+gelu_new) | MLP-out GEMV (+residual)} ->
// ln_f + final_norm + head GEMV.  Every kernel is HBM-bandwidth-shaped: weights are streamed
// exactly once per step as 16-byte-per-lane coalesced loads straight into VGPRs (no LDS round
// trip for a read-once operand), the activation vector lives in LDS, reductions are
// fixed-order wavefront butterflies (bit-reproducible run to run).
//
// Reference arithmetic: indextts/gpt/transformers_gpt2.py:480-667 (block), :1164 (ln_f);
// indextts/gpt/model_v2.py:53,156-160,185 (embed, final_norm+mel_head).
#pragma once
#include "common.h"

namespace ixtts {

constexpr int HD = 64;        // head dim (asserted at create)
constexpr int NSPLIT_MAX = 16;
constexpr int PART_STRIDE = 2 + HD;  // (m, l, acc[64]) per (b, head, split)

enum { IN_LN = 0, IN_LN2 = 1, IN_PLAIN = 2, IN_ATTN = 3 };
enum { EPI_QKV = 0, EPI_RESID = 1, EPI_GELU = 2, EPI_LOGITS = 3, EPI_STORE = 4 };

struct GemvArgs {
  const void* wt;       // [N][K] weights (float or bf16), K contiguous
  const float* bias;    // [N]
  int N;
  int slot0;            // first sequence slot
  // input
  const float* xin;     // IN_LN/IN_LN2: h [slots][D]; IN_PLAIN: [slots][K]; IN_ATTN: partials
  const float* ln_w;    // LN gain/bias (IN_LN, IN_LN2 first norm)
  const float* ln_b;
  const float* ln2_w;   // second norm (IN_LN2)
  const float* ln2_b;
  int nsplit;           // IN_ATTN
  // output
  float* out;           // EPI_RESID: h (in place add); EPI_GELU: ff; EPI_LOGITS: logits; EPI_QKV: q
  int out_stride;       // floats between slots in `out`
  void* kcache;         // EPI_QKV
  void* vcache;
  const int* cur_len;   // [slots] KV position of the token being forwarded
  int layer_stride;     // elements between layers handled by caller (pointer pre-offset); unused
  int smax;             // KV capacity per (slot, head)
  int heads;
  float* norm_out;      // optional: IN_LN2 kernels store the normalised vector (latent rows) [slots][K]
};

// 16 bytes of weights per lane per load, kept RAW in registers until the dot product so
// the loads stay in flight across the LayerNorm / attention-merge prologue.
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

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return __bfloat162float(v); }
__device__ __forceinline__ void store_kv(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_kv(bf16* p, float v) { *p = __float2bfloat16(v); }

__device__ __forceinline__ float gelu_new_f(float x) {
  // 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))   (transformers_gpt2.py:571-585, ACT2FN["gelu_new"])
  const float c = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}

// block-wide sum over 256 threads (4 waves), result broadcast; fixed order
__device__ __forceinline__ float block_sum_256(float v, float* red /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// LayerNorm of xs[0..K) in place (two-pass, eps 1e-5), all 256 threads participate.
template <int K>
__device__ __forceinline__ void layer_norm_lds(float* xs, const float* __restrict__ w, const float* __restrict__ b,
                                               float* red) {
  float s = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) s += xs[i];
  const float mean = block_sum_256(s, red) * (1.0f / K);
  float q = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) {
    float d = xs[i] - mean;
    q += d * d;
  }
  const float var = block_sum_256(q, red) * (1.0f / K);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  for (int i = threadIdx.x; i < K; i += 256) xs[i] = (xs[i] - mean) * rstd * w[i] + b[i];
  __syncthreads();
}

// ------------------------------------------------------------------------------------
// GEMV: y[b][n] = sum_k x[b][k] * Wt[n][k] (+ bias) with fused prologue / epilogue.
// One wave = UNITS units of ROWS consecutive weight rows (ROWS*K elements, contiguous in
// HBM): the wave streams them as ROWS*K/(64*VEC) 16-byte loads per lane.
template <typename WT, int K, int ROWS, int UNITS, int B, int INP, int EPI, typename KVT>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
  constexpr int VEC = WVec<WT>::VEC;
  constexpr int PER = 64 * VEC;            // elements per wave-load
  constexpr int NL = ROWS * K / PER;       // loads per lane per unit
  static_assert(ROWS * K % PER == 0, "unit must be a whole number of wave loads");
  static_assert(K % VEC == 0, "row length must be a multiple of the vector width");
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [B][K]
  __shared__ float red[4];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int unit0 = (blockIdx.x * 4 + wave) * UNITS;
  const int n_units = (a.N + ROWS - 1) / ROWS;

  // ---- issue the weight loads first: they do not depend on the prologue
  uint4 wraw[UNITS][NL];
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int unit = unit0 + u;
    if (unit < n_units) {
      const WT* base = reinterpret_cast<const WT*>(a.wt) + (size_t)unit * ROWS * K;
      const int rows_here = min(ROWS, a.N - unit * ROWS);
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int e = j * PER + lane * VEC;
        if (e < rows_here * K) {
          wraw[u][j] = *reinterpret_cast<const uint4*>(base + e);
        } else {
          wraw[u][j] = make_uint4(0u, 0u, 0u, 0u);
        }
      }
    }
  }

  // ---- prologue: build the activation vector(s) in LDS
#pragma unroll
  for (int b = 0; b < B; ++b) {
    float* x = xs + b * K;
    const int slot = a.slot0 + b;
    if constexpr (INP == IN_LN || INP == IN_LN2) {
      const float* hsrc = a.xin + (size_t)slot * K;
      for (int i = threadIdx.x; i < K; i += 256) x[i] = hsrc[i];
      __syncthreads();
      layer_norm_lds<K>(x, a.ln_w, a.ln_b, red);
      if constexpr (INP == IN_LN2) {
        layer_norm_lds<K>(x, a.ln2_w, a.ln2_b, red);
        if (a.norm_out && blockIdx.x == 0)
          for (int i = threadIdx.x; i < K; i += 256) a.norm_out[(size_t)slot * K + i] = x[i];
      }
    } else if constexpr (INP == IN_PLAIN) {
      const float* src = a.xin + (size_t)slot * K;
      for (int i = threadIdx.x; i < K; i += 256) x[i] = src[i];
    } else {  // IN_ATTN: combine the split-S partials of every head (flash-decode merge)
      const int H = K / HD;
      for (int e = threadIdx.x; e < K; e += 256) {
        const int hh = e / HD, d = e % HD;
        const float* p = a.xin + ((size_t)(slot * H + hh) * a.nsplit) * PART_STRIDE;
        float M = -INFINITY;
        for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, p[s * PART_STRIDE]);
        float l = 0.f, o = 0.f;
        for (int s = 0; s < a.nsplit; ++s) {
          const float ms = p[s * PART_STRIDE];
          if (ms > -INFINITY) {
            const float sc = expf(ms - M);
            l += p[s * PART_STRIDE + 1] * sc;
            o += p[s * PART_STRIDE + 2 + d] * sc;
          }
        }
        x[e] = o / l;
      }
    }
  }
  __syncthreads();

  // ---- dot products
#pragma unroll
  for (int u = 0; u < UNITS; ++u) {
    const int unit = unit0 + u;
    if (unit >= n_units) break;
    float acc[ROWS][B];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int e = j * PER + lane * VEC;
      const int row = e / K;  // K is a compile-time constant
      const int k0 = e - row * K;
      float wv[VEC];
      WVec<WT>::unpack(wraw[u][j], wv);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float* x = xs + b * K + k0;
        float d = 0.f;
#pragma unroll
        for (int v4 = 0; v4 < VEC / 4; ++v4) {
          const float4 xv = *reinterpret_cast<const float4*>(x + v4 * 4);
          d = fmaf(wv[v4 * 4 + 0], xv.x, d);
          d = fmaf(wv[v4 * 4 + 1], xv.y, d);
          d = fmaf(wv[v4 * 4 + 2], xv.z, d);
          d = fmaf(wv[v4 * 4 + 3], xv.w, d);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r][b] += (row == r) ? d : 0.f;
      }
    }
    // ---- reduce + epilogue (lane 0 of the wave owns the unit's rows)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = unit * ROWS + r;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float tot = wave_sum(acc[r][b]);
        if (lane == 0 && n < a.N) {
          const int slot = a.slot0 + b;
          float v = tot + a.bias[n];
          if constexpr (EPI == EPI_RESID) {
            float* o = a.out + (size_t)slot * a.out_stride + n;
            *o = *o + v;
          } else if constexpr (EPI == EPI_GELU) {
            a.out[(size_t)slot * a.out_stride + n] = gelu_new_f(v);
          } else if constexpr (EPI == EPI_LOGITS || EPI == EPI_STORE) {
            a.out[(size_t)slot * a.out_stride + n] = v;
          } else {  // EPI_QKV: q -> buffer, k/v -> cache at position cur_len[slot]
            constexpr int D = K;
            if (n < D) {
              a.out[(size_t)slot * a.out_stride + n] = v;
            } else {
              const int which = n / D;  // 1: k, 2: v
              const int c = n - which * D;
              const int hh = c / HD, d = c % HD;
              const int pos = a.cur_len[slot];
              KVT* cache = reinterpret_cast<KVT*>(which == 1 ? a.kcache : a.vcache);
              store_kv(cache + (((size_t)slot * a.heads + hh) * a.smax + pos) * HD + d, v);
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Split-S single-query attention (flash-decode): grid (H, nsplit, B), 256 threads.
// lane -> (position lane>>2 of a 16-position group, 16-dim slice lane&3); each 4-lane
// group keeps an online-softmax state over its positions; states are merged with
// wavefront shuffles, then across the 4 waves through LDS.
struct AttnArgs {
  const float* q;       // [slots][D]
  const void* kcache;   // layer base: [slots][H][smax][64]
  const void* vcache;
  float* part;          // [slots][H][nsplit][PART_STRIDE]
  const int* cur_len;   // position of the new token; keys [valid_from, cur_len]
  const int* valid_from;
  int slot0, heads, smax, nsplit, D;
};

template <typename KVT>
__device__ __forceinline__ void load16(const KVT* p, float (&o)[16]);
template <>
__device__ __forceinline__ void load16<float>(const float* p, float (&o)[16]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 v = reinterpret_cast<const float4*>(p)[i];
    o[4 * i] = v.x; o[4 * i + 1] = v.y; o[4 * i + 2] = v.z; o[4 * i + 3] = v.w;
  }
}
template <>
__device__ __forceinline__ void load16<bf16>(const bf16* p, float (&o)[16]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint4 v = reinterpret_cast<const uint4*>(p)[i];
    o[8 * i + 0] = lo_bf16(v.x); o[8 * i + 1] = hi_bf16(v.x); o[8 * i + 2] = lo_bf16(v.y); o[8 * i + 3] = hi_bf16(v.y);
    o[8 * i + 4] = lo_bf16(v.z); o[8 * i + 5] = hi_bf16(v.z); o[8 * i + 6] = lo_bf16(v.w); o[8 * i + 7] = hi_bf16(v.w);
  }
}

template <typename KVT>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a) {
  __shared__ float sm[4][4][2 + 16];  // [wave][dpart][m,l,acc16]
  const int hh = blockIdx.x, split = blockIdx.y, slot = a.slot0 + blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pg = lane >> 2, dp = lane & 3;
  const int S = a.cur_len[slot] + 1;
  const int vf = a.valid_from[slot];
  const int n = S - vf;
  const int chunk = (n + a.nsplit - 1) / a.nsplit;
  const int p_begin = vf + split * chunk;
  const int p_end = min(S, p_begin + chunk);

  float qv[16];
  {
    const float* qp = a.q + (size_t)slot * a.D + hh * HD + dp * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) qv[i] = qp[i] * 0.125f;  // 1/sqrt(64)
  }
  const KVT* kb = reinterpret_cast<const KVT*>(a.kcache) + ((size_t)slot * a.heads + hh) * a.smax * HD + dp * 16;
  const KVT* vb = reinterpret_cast<const KVT*>(a.vcache) + ((size_t)slot * a.heads + hh) * a.smax * HD + dp * 16;

  float m = -INFINITY, l = 0.f, acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  for (int p0 = p_begin + wave * 16; p0 < p_end; p0 += 64) {
    const int p = p0 + pg;
    const bool ok = p < p_end;
    float kv[16], vv[16];
    float s = 0.f;
    if (ok) {
      load16<KVT>(kb + (size_t)p * HD, kv);
      load16<KVT>(vb + (size_t)p * HD, vv);
#pragma unroll
      for (int i = 0; i < 16; ++i) s = fmaf(qv[i], kv[i], s);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (ok) {
      const float mn = fmaxf(m, s);
      const float sc = expf(m - mn);  // exp(-inf) = 0 on the first hit
      const float pw = expf(s - mn);
      l = l * sc + pw;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(acc[i], sc, pw * vv[i]);
      m = mn;
    }
  }
  // merge the 16 position groups of the wave (lanes with equal dp): xor 4, 8, 16, 32
#pragma unroll
  for (int o = 4; o <= 32; o <<= 1) {
    const float m2 = __shfl_xor(m, o, 64);
    const float l2 = __shfl_xor(l, o, 64);
    const float mn = fmaxf(m, m2);
    const float s1 = (m > -INFINITY) ? expf(m - mn) : 0.f;
    const float s2 = (m2 > -INFINITY) ? expf(m2 - mn) : 0.f;
    l = l * s1 + l2 * s2;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float a2 = __shfl_xor(acc[i], o, 64);
      acc[i] = acc[i] * s1 + a2 * s2;
    }
    m = mn;
  }
  if (pg == 0) {
    sm[wave][dp][0] = m;
    sm[wave][dp][1] = l;
#pragma unroll
    for (int i = 0; i < 16; ++i) sm[wave][dp][2 + i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    // thread t -> output dim t: merge 4 waves in fixed order
    const int d = threadIdx.x, dpp = d >> 4, di = d & 15;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, sm[w][dpp][0]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = sm[w][dpp][0];
      if (mw > -INFINITY) {
        const float sc = expf(mw - M);
        L += sm[w][dpp][1] * sc;
        O += sm[w][dpp][2 + di] * sc;
      }
    }
    float* out = a.part + (((size_t)slot * a.heads + hh) * a.nsplit + split) * PART_STRIDE;
    if (d == 0) {
      out[0] = M;
      out[1] = L;
    }
    out[2 + d] = O;
  }
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
  int V, D, max_new, n_pos, stop, slot0;
};

__global__ __launch_bounds__(1024) void sampler_greedy_kernel(SamplerState s) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  __shared__ int tok_s;
  const int slot = s.slot0 + blockIdx.x;
  const float theta = s.cfg->repetition_penalty;
  const int suppress = s.cfg->suppress_stop;
  const float* lg = s.logits + (size_t)slot * s.V;
  const uint8_t* seen = s.seen + (size_t)slot * s.V;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int v = threadIdx.x; v < s.V; v += 1024) {
    float x = lg[v];
    if (suppress && v == s.stop) x = -INFINITY;
    if (seen[v]) x = (x < 0.f) ? x * theta : x / theta;  // RepetitionPenaltyLogitsProcessor
    if (x > best || (x == best && v < besti)) {
      best = x;
      besti = v;
    }
  }
  // argmax with lowest-index tie break (torch.argmax returns the first maximal element)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) {
      best = ob;
      besti = oi;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    bv[threadIdx.x >> 6] = best;
    bi[threadIdx.x >> 6] = besti;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < besti)) {
        best = bv[w];
        besti = bi[w];
      }
    int tok = besti;
    if (s.finished[slot]) tok = s.stop;  // finished rows keep emitting pad == stop (generation_utils.py:3255-3256)
    if (s.forced[slot] >= 0) {
      tok = s.forced[slot];
      s.forced[slot] = -1;
    }
    const int k = s.gen_count[slot] + 1;  // this is the k-th generated token
    if (k <= s.max_new) s.tokens[(size_t)slot * s.max_new + k - 1] = tok;
    s.seen[(size_t)slot * s.V + tok] = 1;
    if (tok == s.stop) s.finished[slot] = 1;
    s.gen_count[slot] = k;
    s.cur_len[slot] = s.prompt_len[slot] + k - 1;
    tok_s = tok;
  }
  __syncthreads();
  // embed: mel_embedding[tok] + mel_pos_embedding[k + 1]   (model_v2.py:156-160)
  const int tok = tok_s;
  const int k = s.gen_count[slot];
  const int pos = min(k + 1, s.n_pos - 1);
  const float* e = s.mel_emb + (size_t)tok * s.D;
  const float* pe = s.mel_pos + (size_t)pos * s.D;
  float* h = s.h + (size_t)slot * s.D;
  for (int i = threadIdx.x; i < s.D; i += 1024) h[i] = e[i] + pe[i];
}

// copy one embedding row into the residual stream of `slot` and set its KV position
__global__ void set_row_kernel(float* h, const float* row, const float* add, int D, int slot, int* cur_len, int pos) {
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < D; i += blockDim.x * gridDim.x)
    h[(size_t)slot * D + i] = row[i] + (add ? add[i] : 0.f);
  if (blockIdx.x == 0 && threadIdx.x == 0) cur_len[slot] = pos;
}

// ---- weight packing (device side): [K][N] fp32 -> Wt[N][K] in WT ; or plain convert
template <typename WT>
__global__ void pack_transpose_kernel(const float* __restrict__ src, WT* __restrict__ dst, int K, int N) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    int k = k0 + i, n = n0 + threadIdx.x;
    tile[i][threadIdx.x] = (k < K && n < N) ? src[(size_t)k * N + n] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    int n = n0 + i, k = k0 + threadIdx.x;
    if (n < N && k < K) {
      if constexpr (sizeof(WT) == 4) dst[(size_t)n * K + k] = tile[threadIdx.x][i];
      else dst[(size_t)n * K + k] = __float2bfloat16(tile[threadIdx.x][i]);
    }
  }
}
template <typename WT>
__global__ void pack_convert_kernel(const float* __restrict__ src, WT* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (sizeof(WT) == 4) dst[i] = src[i];
    else dst[i] = __float2bfloat16(src[i]);
  }
}

}  // namespace ixtts
