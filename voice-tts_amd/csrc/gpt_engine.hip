// gpt_engine.hip -- autoregressive GPT-2 decode engine on gfx950: host side + C ABI (seam 1).
//
// Replaces the object DeepSpeed swaps in for `UnifiedVoice.inference_model`
// (indextts/gpt/model_v2.py:433-446).  Reference behaviour restated:
//   prefill / decode embed   model_v2.py:144-160 (F6: k-th generated token sits at mel-pos k+1)
//   trunk                    transformers_gpt2.py:480-667,985-1184 (wpe == 0, model_v2.py:22-23,274)
//   head                     model_v2.py:53,185 (final_norm after ln_f, then mel_head)
//   token selection          transformers_generation_utils.py:3196-3269 + RepetitionPenalty
//   latent pass              model_v2.py:554-596,486-512
//
// HBM layout (one arena, broadcastable with one RCCL call):
//   per layer: ln_1 w,b | Wqkv^T [3D][D] | b | Wo^T [D][D] | b | ln_2 w,b | Wfc^T [4D][D] | b | Wpr^T [D][4D] | b
//   ln_f, final_norm, Whead [V][D], b, mel_embedding [V][D] fp32, mel_pos_embedding [n_pos][D] fp32.
//   Matrices are stored TRANSPOSED (output row contiguous over K) in fp32 or bf16 so one
//   wavefront streams whole rows with 16-byte-per-lane loads; vectors stay fp32.
//   KV cache: [layer][slot][head][max_seq][64] (fp32 in parity mode, bf16 in throughput mode).
#define IXTTS_ENGINE_TU 1
#include <type_traits>

#include "gpt_engine.h"
#include "gpt_kernels.h"
#include "gpt_wide.h"

using namespace ixtts;

// ------------------------------------------------------------------------------------ launch helpers
#ifdef IXTTS_TRACE
static int g_trace_seq = 0;  // one id per traced launch (baked into captured graphs)
#define IXTTS_TRACE_ARG , g_trace_seq++
#else
#define IXTTS_TRACE_ARG
#endif
template <typename WT, typename KVT, int K, int ROWS, int UNITS, int B, int INP, int EPI, int WPB = 4, bool XLDS = false>
static int launch_gemv(const GemvArgs& a, hipStream_t st) {
  auto kern = gemv_reg_kernel<WT, K, ROWS, UNITS, B, INP, EPI, KVT, WPB, XLDS>;
  const int n_units = (a.N + ROWS - 1) / ROWS;
  const int grid = ceil_div(n_units, WPB * UNITS);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPB), 0, st, a.wt, a.xin, a.bias, a.out, a.N, a.slot0, a.out_stride, a.smax, a.kcache, a.vcache,
                     a.cur_len, a.heads, a.nsplit, a.ln_w, a.ln_b, a.aux, a.xout IXTTS_PF_ARGS(a) IXTTS_TRACE_ARG);
  return IXTTS_OK;
}

// IXTTS_PF builds: `a` also fetches ahead for the launch that reads matrix `next` next: which 0 qkv, 1 attn-out, 2 fc, 3 mlp-out of
// layer l, 4 head; -1: nothing (its own matrix again: L2 hits)
template <int D>
static void set_prefetch(ixtts_gpt* h, GemvArgs& a, int which, int l) {
  using DM = Dims<D>;
  const size_t es = h->esize;
  a.pf_ptr = a.wt;
  a.pf_stride = 1024;
  a.pf_per = 1;
  a.pf_nwaves = 1;
  if (which < 0 || h->pf_per <= 0) return;
  size_t off = 0;
  int rows_per_wave = 0, K = D, N = 0;
  if (which == 4) {
    off = h->whead; rows_per_wave = DM::R1 * DM::U_HEAD; N = h->V;
  } else {
    const LayerOff& o = h->lo[l];
    switch (which) {
      case 0: off = o.wqkv; rows_per_wave = DM::R1 * DM::U_QKV; N = 3 * D; break;
      case 1: off = o.wo; rows_per_wave = DM::R1 * DM::U_OUT; N = D; break;
      case 2: off = o.wfc; rows_per_wave = DM::R1 * DM::U_FC; N = 4 * D; break;
      default: off = o.wpr; rows_per_wave = DM::R4 * DM::U_PR; K = 4 * D; N = D; break;
    }
  }
  a.pf_ptr = A_PTR(off);
  a.pf_stride = (int)(rows_per_wave * K * es);
  a.pf_nwaves = N / rows_per_wave;  // (whole waves only: a ragged last wave is not fetched ahead)
  a.pf_per = std::min(h->pf_per, a.pf_stride / 1024);
}

template <typename WT, typename KVT, int K, int ROWS, int UNITS, int B, int EPI, int WPB = 4>
static int launch_gemv_lds(const GemvArgs& a, hipStream_t st) {
  auto kern = gemv_lds_kernel<WT, K, ROWS, UNITS, B, EPI, KVT, WPB>;
  const size_t smem = (size_t)B * K * sizeof(float);
  if (smem > 64 * 1024) {
    static bool done = false;
    if (!done) {
      IX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      done = true;
    }
  }
  const int n_units = (a.N + ROWS - 1) / ROWS;
  const int grid = ceil_div(n_units, WPB * UNITS);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPB), smem, st, a.wt, a.xin, a.bias, a.out, a.N, a.slot0, a.out_stride, a.smax, a.kcache,
                     a.vcache, a.cur_len, a.heads, a.nsplit, a.ln_w, a.ln_b, a.aux, a.xout IXTTS_PF_ARGS(a) IXTTS_TRACE_ARG);
  return IXTTS_OK;
}

template <typename WT, typename KVT, int D, int B>
static int gemv_qkv(ixtts_gpt* h, int l, int slot0, hipStream_t st) {
  using DM = Dims<D>;
  const LayerOff& o = h->lo[l];
  GemvArgs a;
  memset(&a, 0, sizeof(a));
  a.wt = A_PTR(o.wqkv);
  a.bias = A_F32(o.bqkv);
  a.N = 3 * D;
  a.slot0 = slot0;
  a.xin = h->hc;
  a.out = h->q;
  a.out_stride = D;
  const size_t lstride = (size_t)h->slots * D * h->smax * sizeof(KVT);
  a.kcache = (uint8_t*)h->kc + l * lstride;
  a.vcache = (uint8_t*)h->vc + l * lstride;
  a.cur_len = h->cur_len;
  a.smax = h->smax;
  a.heads = h->H;
  set_prefetch<D>(h, a, 1, l);  // the attention launch in between reads no weights
  if constexpr (std::is_same<WT, bf16>::value && D == MLP_D) {
    if (h->mlp_fused) {
      if (l + 1 < h->L) a.aux = h->mlp_ctr + (size_t)l * MLP_CTR_STRIDE;  // this layer's MLP is fused: clear its counters
      if (l > 0) {  // the previous layer's MLP left 8 partial sums: complete the residual stream into the other buffer
        a.ln_w = h->mlp_part;
        a.ln_b = A_F32(h->lo[l - 1].bpr);
        a.nsplit = h->slots;
        a.xout = (h->hc == h->h) ? h->h2 : h->h;
        const int rc = launch_gemv<WT, KVT, D, DM::R1, DM::U_QKV, B, IN_LN_PART, EPI_QKV, DM::W_QKV, true>(a, st);
        h->hc = a.xout;
        return rc;
      }
    }
  }
  return launch_gemv<WT, KVT, D, DM::R1, DM::U_QKV, B, IN_LN, EPI_QKV, DM::W_QKV, DM::XLDS>(a, st);
}

template <typename WT, typename KVT, int D, int B>
static int gemv_out(ixtts_gpt* h, int l, int slot0, hipStream_t st) {
  using DM = Dims<D>;
  const LayerOff& o = h->lo[l];
  GemvArgs a;
  memset(&a, 0, sizeof(a));
  a.wt = A_PTR(o.wo);
  a.bias = A_F32(o.bo);
  a.N = D;
  a.slot0 = slot0;
  a.out = h->hc;
  a.out_stride = D;
  set_prefetch<D>(h, a, 2, l);
  if (h->attn_bucket < NBKT) {  // merge the split-S partials while staging
    static_assert(ATTN_NSP == 4 && NBKT == 8, "merge variant / bucket switch above");
    a.xin = h->part;
    return launch_gemv<WT, KVT, D, DM::R1, DM::U_OUT, B, IN_ATTN4, EPI_RESID, 4, true>(a, st);
  }
  a.xin = h->att;
  return launch_gemv<WT, KVT, D, DM::R1, DM::U_OUT, B, IN_PLAIN, EPI_RESID>(a, st);
}

template <typename WT, typename KVT, int D, int B>
static int gemv_fc(ixtts_gpt* h, int l, int slot0, hipStream_t st) {
  using DM = Dims<D>;
  const LayerOff& o = h->lo[l];
  GemvArgs a;
  memset(&a, 0, sizeof(a));
  a.wt = A_PTR(o.wfc);
  a.bias = A_F32(o.bfc);
  a.N = 4 * D;
  a.slot0 = slot0;
  a.xin = h->hc;
  a.out = h->ff;
  a.out_stride = 4 * D;
  set_prefetch<D>(h, a, 3, l);
  return launch_gemv<WT, KVT, D, DM::R1, DM::U_FC, B, IN_LN, EPI_GELU, DM::W_FC, DM::XLDS>(a, st);
}

template <typename WT, typename KVT, int D, int B>
static int gemv_pr(ixtts_gpt* h, int l, int slot0, hipStream_t st) {
  using DM = Dims<D>;
  const LayerOff& o = h->lo[l];
  GemvArgs a;
  memset(&a, 0, sizeof(a));
  a.wt = A_PTR(o.wpr);
  a.bias = A_F32(o.bpr);
  a.N = D;
  a.slot0 = slot0;
  a.xin = h->ff;
  a.out = h->hc;
  a.out_stride = D;
  set_prefetch<D>(h, a, l + 1 < h->L ? 0 : 4, l + 1);
  return launch_gemv_lds<WT, KVT, 4 * D, DM::R4, DM::U_PR, B, EPI_RESID, DM::W_PR>(a, st);
}

template <typename WT, typename KVT, int D, int B>
static int gemv_head(ixtts_gpt* h, int slot0, float* norm_out, hipStream_t st) {
  using DM = Dims<D>;
  GemvArgs a;
  memset(&a, 0, sizeof(a));
  a.wt = A_PTR(h->whead);
  a.bias = A_F32(h->bhead);
  a.N = h->V;
  a.slot0 = slot0;
  a.xin = h->hc;
  a.ln_w = A_F32(h->lnf_w);
  a.ln_b = A_F32(h->lnf_b);
  a.out = h->logits;
  a.out_stride = h->V;
  a.norm_out = norm_out;
  set_prefetch<D>(h, a, 0, 0);  // the next step's first matrix (the sampler launch in between reads no weights)
  return launch_gemv<WT, KVT, D, DM::R1, DM::U_HEAD, B, IN_LN2, EPI_LOGITS, 4, true>(a, st);
}

template <typename WT, typename KVT, int D, int B>
static int forward_layers(ixtts_gpt* h, int slot0, hipStream_t st) {
  const size_t lstride = (size_t)h->slots * D * h->smax * sizeof(KVT);
  h->hc = h->h;  // the sampler / prefill leave the token's embedding here; fused-MLP layers alternate between h and h2
  for (int l = 0; l < h->L; ++l) {
    const void* kcl = (uint8_t*)h->kc + l * lstride;
    const void* vcl = (uint8_t*)h->vc + l * lstride;
    IX_TRY((gemv_qkv<WT, KVT, D, B>(h, l, slot0, st)));
    if (h->attn_bucket < NBKT) {
      constexpr int F = 8 / KVLayout<KVT>::PPW;  // fp32 cache: half the keys per wave-load, twice the blocks
      const dim3 grid(h->H, ATTN_NSP, B);
#define IX_ATTN_SPLIT(IT) \
  hipLaunchKernelGGL((attn_split_kernel<KVT, IT * F, ATTN_NSP>), grid, dim3(256), 0, st, h->q, kcl, vcl, h->cur_len, h->valid_from, h->smax, h->H, slot0, D, \
                     h->part IXTTS_TRACE_ARG)
      switch (h->attn_bucket) {
        case 0: IX_ATTN_SPLIT(ATTN_IT[0]); break;
        case 1: IX_ATTN_SPLIT(ATTN_IT[1]); break;
        case 2: IX_ATTN_SPLIT(ATTN_IT[2]); break;
        case 3: IX_ATTN_SPLIT(ATTN_IT[3]); break;
        case 4: IX_ATTN_SPLIT(ATTN_IT[4]); break;
        case 5: IX_ATTN_SPLIT(ATTN_IT[5]); break;
        case 6: IX_ATTN_SPLIT(ATTN_IT[6]); break;
        default: IX_ATTN_SPLIT(ATTN_IT[7]); break;
      }
#undef IX_ATTN_SPLIT
    } else {
      // any context length: one workgroup per (head, slot), online softmax over as many passes as it takes
      hipLaunchKernelGGL((attn_decode_kernel<KVT, 8, 4>), dim3(h->H, 1, B), dim3(512), 0, st, h->q, kcl, vcl, h->cur_len, h->valid_from, h->smax,
                         h->H, slot0, D, h->att, 1 IXTTS_TRACE_ARG);
    }
    IX_TRY((gemv_out<WT, KVT, D, B>(h, l, slot0, st)));
    if constexpr (std::is_same<WT, bf16>::value && D == MLP_D) {
      if (h->mlp_fused && l + 1 < h->L) {  // (the last layer stays split: the head reads a finished residual stream)
        const LayerOff& o = h->lo[l];
        hipLaunchKernelGGL((mlp_fused_kernel<B>), dim3(256), dim3(64 * MLP_WAVES), 0, st, reinterpret_cast<const bf16*>(A_PTR(o.wfc)), (const float*)h->hc,
                           (const float*)A_F32(o.bfc), h->ff, reinterpret_cast<const bf16*>(h->wprx) + (size_t)l * MLP_D * MLP_FF, h->mlp_part,
                           h->mlp_ctr + (size_t)l * MLP_CTR_STRIDE, h->mlp_ctr + (size_t)h->L * MLP_CTR_STRIDE, slot0, h->slots);
        continue;
      }
    }
    IX_TRY((gemv_fc<WT, KVT, D, B>(h, l, slot0, st)));
    IX_TRY((gemv_pr<WT, KVT, D, B>(h, l, slot0, st)));
  }
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

static SamplerState make_sampler_state(ixtts_gpt* h) {
  SamplerState s;
  s.logits = h->logits;
  s.seen = h->seen;
  s.tokens = h->tokens;
  s.gen_count = h->gen_count;
  s.cur_len = h->cur_len;
  s.prompt_len = h->prompt_len;
  s.finished = h->finished;
  s.forced = h->forced;
  s.h = h->h;
  s.mel_emb = A_F32(h->mel_emb);
  s.mel_pos = A_F32(h->mel_pos);
  s.cfg = h->d_samp;
  s.V = h->V;
  s.D = h->D;
  s.max_new = h->smax;
  s.n_pos = h->cfg.n_mel_pos;
  s.stop = h->cfg.stop_mel_token;
  s.slot0 = 0;
  s.probs_out = h->probs;
  return s;
}

static void launch_sampler(ixtts_gpt* h, int n_active, hipStream_t st) {
  SamplerState s = make_sampler_state(h);
  hipLaunchKernelGGL(sampler_kernel, dim3(n_active), dim3(1024), 0, st, s);
}

// dispatch on (dtype, D, B)
#define DISPATCH_B(FN, WT, KVT, DD, B, ...)                 \
  switch (B) {                                               \
    case 1: return FN<WT, KVT, DD, 1>(__VA_ARGS__);          \
    case 2: return FN<WT, KVT, DD, 2>(__VA_ARGS__);          \
    case 3: return FN<WT, KVT, DD, 3>(__VA_ARGS__);          \
    case 4: return FN<WT, KVT, DD, 4>(__VA_ARGS__);          \
    default: set_error("batch %d unsupported on the register GEMVs (1..4)", B); return IXTTS_ERR_ARG; \
  }
#define DISPATCH(FN, h, B, ...)                                                              \
  do {                                                                                       \
    if ((h)->cfg.weight_dtype == IXTTS_DTYPE_F32) {                                          \
      if ((h)->D == 1280) { DISPATCH_B(FN, float, float, 1280, B, __VA_ARGS__) }             \
      else { DISPATCH_B(FN, float, float, 128, B, __VA_ARGS__) }                             \
    } else {                                                                                 \
      if ((h)->D == 1280) { DISPATCH_B(FN, bf16, bf16, 1280, B, __VA_ARGS__) }               \
      else { DISPATCH_B(FN, bf16, bf16, 128, B, __VA_ARGS__) }                               \
    }                                                                                        \
  } while (0)


// ------------------------------------------------------------------------------------ wide engines (gpt_wide.h)
// max_batch > 4: bf16 only.  Rows per workgroup (RP0 + RP1) make one workgroup per CU at model_dim 1280: 15 / 5 / 16 + 4 / 5 / 16 + 16.
template <int K, int INP, int EPI, int NW, int RP0, int RP1>
static int launch_wide(const void* wt, const void* xin, const float* bias, void* out, int N, int B, int slot0, int out_stride, ixtts_gpt* h,
                       void* kc, void* vc, const float* ln_w, const float* ln_b, hipStream_t st) {
  constexpr int NT = RP1 > 0 ? 2 : 1;
  hipLaunchKernelGGL((gemv_wide_kernel<K, NT, INP, EPI, bf16, NW, RP0, RP1>), dim3(ceil_div(N, RP0 + RP1)), dim3(64 * NW), 0, st,
                     reinterpret_cast<const bf16*>(wt), xin, bias, out, N, B, slot0, out_stride, h->smax, kc, vc, (const int*)h->cur_len, h->H, ln_w, ln_b);
  return IXTTS_OK;
}

template <int D>
static int wide_which(ixtts_gpt* h, int which, int l, int B, int slot0, hipStream_t st) {
  const LayerOff& o = h->lo[std::min(l, h->L - 1)];
  const size_t lstride = (size_t)h->slots * D * h->smax * sizeof(bf16);
  switch (which) {
    case 0:
      return launch_wide<D, WIN_LN, EPI_QKV, 4, 15, 0>(A_PTR(o.wqkv), h->h, A_F32(o.bqkv), h->q, 3 * D, B, slot0, D, h, (uint8_t*)h->kc + l * lstride,
                                                 (uint8_t*)h->vc + l * lstride, nullptr, nullptr, st);
    case 1:
      return launch_wide<D, WIN_PLAIN, EPI_RESID, 4, 5, 0>(A_PTR(o.wo), h->att, A_F32(o.bo), h->h, D, B, slot0, D, h, nullptr, nullptr, nullptr, nullptr, st);
    case 2:
      return launch_wide<D, WIN_LN, EPI_GELU, 4, 16, 4>(A_PTR(o.wfc), h->h, A_F32(o.bfc), h->ff, 4 * D, B, slot0, 4 * D, h, nullptr, nullptr, nullptr, nullptr, st);
    case 3:
      return launch_wide<4 * D, WIN_FF, EPI_RESID, 4, 5, 0>(A_PTR(o.wpr), h->ff, A_F32(o.bpr), h->h, D, B, slot0, D, h, nullptr, nullptr, nullptr, nullptr, st);
    case 4:
      return launch_wide<D, WIN_LN2, EPI_LOGITS, 4, 16, 16>(A_PTR(h->whead), h->h, A_F32(h->bhead), h->logits, h->V, B, slot0, h->V, h, nullptr, nullptr,
                                                    A_F32(h->lnf_w), A_F32(h->lnf_b), st);
  }
  set_error("wide_which: %d", which);
  return IXTTS_ERR_ARG;
}

// sampler / prefill leave the token's embedding in h->h; one workgroup per (head, slot) sweeps the whole context (the split-S
// partials would have to be re-read by every out-proj workgroup: 21 KB per sequence)
template <int D>
static int forward_layers_wide(ixtts_gpt* h, int B, int slot0, hipStream_t st) {
  const size_t lstride = (size_t)h->slots * D * h->smax * sizeof(bf16);
  h->hc = h->h;
  for (int l = 0; l < h->L; ++l) {
    IX_TRY(wide_which<D>(h, 0, l, B, slot0, st));
    hipLaunchKernelGGL((attn_decode_kernel<bf16, 8, 4>), dim3(h->H, 1, B), dim3(512), 0, st, h->q, (const void*)((uint8_t*)h->kc + l * lstride),
                       (const void*)((uint8_t*)h->vc + l * lstride), h->cur_len, h->valid_from, h->smax, h->H, slot0, D, h->att, 1 IXTTS_TRACE_ARG);
    IX_TRY(wide_which<D>(h, 1, l, B, slot0, st));
    IX_TRY(wide_which<D>(h, 2, l, B, slot0, st));
    IX_TRY(wide_which<D>(h, 3, l, B, slot0, st));
  }
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

static int do_forward_layers(ixtts_gpt* h, int B, int slot0, hipStream_t st) {
  if (h->wide) return h->D == 1280 ? forward_layers_wide<1280>(h, B, slot0, st) : forward_layers_wide<128>(h, B, slot0, st);
  DISPATCH(forward_layers, h, B, h, slot0, st);
}
static int do_head(ixtts_gpt* h, int B, int slot0, float* norm_out, hipStream_t st) {
  if (h->wide) return h->D == 1280 ? wide_which<1280>(h, 4, 0, B, slot0, st) : wide_which<128>(h, 4, 0, B, slot0, st);
  DISPATCH(gemv_head, h, B, h, slot0, norm_out, st);
}
static int do_gemv_which(ixtts_gpt* h, int which, int l, int B, hipStream_t st) {
  if (h->wide) return h->D == 1280 ? wide_which<1280>(h, which, l, B, 0, st) : wide_which<128>(h, which, l, B, 0, st);
  switch (which) {
    case 0: DISPATCH(gemv_qkv, h, B, h, l, 0, st);
    case 1: DISPATCH(gemv_out, h, B, h, l, 0, st);
    case 2: DISPATCH(gemv_fc, h, B, h, l, 0, st);
    case 3: DISPATCH(gemv_pr, h, B, h, l, 0, st);
    case 4: DISPATCH(gemv_head, h, B, h, 0, nullptr, st);
  }
  set_error("bench_gemv: which=%d", which);
  return IXTTS_ERR_ARG;
}

// ------------------------------------------------------------------------------------ create
static size_t take(size_t& off, size_t bytes) {
  size_t o = off;
  off = align_up(off + bytes, 256);
  return o;
}

extern "C" int ixtts_gpt_create(ixtts_gpt** out, const ixtts_gpt_cfg* c) {
  IX_ARG(out && c, "gpt_create: null argument");
  IX_ARG(c->model_dim == 1280 || c->model_dim == 128, "gpt_create: model_dim %d has no kernel instantiation (1280 | 128)", c->model_dim);
  IX_ARG(c->heads * HD == c->model_dim, "gpt_create: head dim must be 64 (heads %d, dim %d)", c->heads, c->model_dim);
  IX_ARG(c->layers > 0 && c->n_mel_codes > 0 && c->n_mel_pos > 2 && c->max_seq > 8, "gpt_create: bad sizes");
  IX_ARG(c->max_batch >= 1 && c->max_batch <= MAXB, "gpt_create: max_batch %d (1..%d)", c->max_batch, MAXB);
  IX_ARG(c->weight_dtype == IXTTS_DTYPE_F32 || c->weight_dtype == IXTTS_DTYPE_BF16, "gpt_create: weight_dtype");
  IX_ARG(c->max_batch <= MAXB_REG || c->weight_dtype == IXTTS_DTYPE_BF16, "gpt_create: max_batch %d > %d runs on the bf16 matrix cores: needs bf16 weights", c->max_batch, MAXB_REG);
  IX_ARG(c->start_mel_token >= 0 && c->start_mel_token < c->n_mel_codes && c->stop_mel_token >= 0 && c->stop_mel_token < c->n_mel_codes, "gpt_create: start/stop token out of range");
  auto* h = new (std::nothrow) ixtts_gpt();
  if (!h) return IXTTS_ERR_NOMEM;
  h->cfg = *c;
  h->D = c->model_dim;
  h->L = c->layers;
  h->H = c->heads;
  h->V = c->n_mel_codes;
  h->FF = 4 * h->D;
  h->slots = c->max_batch + 1;  // last slot: scratch sequence for the latent pass
  h->smax = c->max_seq;
  h->esize = c->weight_dtype == IXTTS_DTYPE_F32 ? 4 : 2;
  const int D = h->D, FF = h->FF, V = h->V;
  const size_t es = h->esize;
  size_t off = 0;
  size_t stage_off = 0;
  auto reg = [&](const std::string& name, size_t o, int kind, int64_t d0, int64_t d1) {
    TDesc t;
    t.off = o;
    t.kind = kind;
    t.d0 = d0;
    t.d1 = d1;
    if (kind == T_MAT_T || kind == T_MAT_N) {
      t.stage_off = stage_off;
      stage_off += (size_t)d0 * d1;
    }
    h->tens[name] = t;
  };
  h->lo.resize(h->L);
  for (int l = 0; l < h->L; ++l) {
    LayerOff& o = h->lo[l];
    std::string p = "gpt.h." + std::to_string(l) + ".";
    o.ln1_w = take(off, D * 4); reg(p + "ln_1.weight", o.ln1_w, T_VEC, D, 0);
    o.ln1_b = take(off, D * 4); reg(p + "ln_1.bias", o.ln1_b, T_VEC, D, 0);
    o.wqkv = take(off, (size_t)3 * D * D * es); reg(p + "attn.c_attn.weight", o.wqkv, T_MAT_T, D, 3 * D);
    o.bqkv = take(off, 3 * D * 4); reg(p + "attn.c_attn.bias", o.bqkv, T_VEC, 3 * D, 0);
    o.wo = take(off, (size_t)D * D * es); reg(p + "attn.c_proj.weight", o.wo, T_MAT_T, D, D);
    o.bo = take(off, D * 4); reg(p + "attn.c_proj.bias", o.bo, T_VEC, D, 0);
    o.ln2_w = take(off, D * 4); reg(p + "ln_2.weight", o.ln2_w, T_VEC, D, 0);
    o.ln2_b = take(off, D * 4); reg(p + "ln_2.bias", o.ln2_b, T_VEC, D, 0);
    o.wfc = take(off, (size_t)FF * D * es); reg(p + "mlp.c_fc.weight", o.wfc, T_MAT_T, D, FF);
    o.bfc = take(off, FF * 4); reg(p + "mlp.c_fc.bias", o.bfc, T_VEC, FF, 0);
    o.wpr = take(off, (size_t)D * FF * es); reg(p + "mlp.c_proj.weight", o.wpr, T_MAT_T, FF, D);
    o.bpr = take(off, D * 4); reg(p + "mlp.c_proj.bias", o.bpr, T_VEC, D, 0);
  }
  h->lnf_w = take(off, D * 4); reg("gpt.ln_f.weight", h->lnf_w, T_VEC, D, 0);
  h->lnf_b = take(off, D * 4); reg("gpt.ln_f.bias", h->lnf_b, T_VEC, D, 0);
  h->fn_w = take(off, D * 4); reg("final_norm.weight", h->fn_w, T_VEC, D, 0);
  h->fn_b = take(off, D * 4); reg("final_norm.bias", h->fn_b, T_VEC, D, 0);
  h->whead = take(off, (size_t)V * D * es); reg("mel_head.weight", h->whead, T_MAT_N, V, D);
  h->bhead = take(off, V * 4); reg("mel_head.bias", h->bhead, T_VEC, V, 0);
  h->mel_emb = take(off, (size_t)V * D * 4); reg("mel_embedding.weight", h->mel_emb, T_EMB, V, D);
  h->mel_pos = take(off, (size_t)c->n_mel_pos * D * 4); reg("mel_pos_embedding.emb.weight", h->mel_pos, T_EMB, c->n_mel_pos, D);
  h->arena_bytes = off;
  h->stage_floats = stage_off;

  auto fail = [&](const char* what) {
    set_error("gpt_create: allocation failed (%s): %s", what, hipGetErrorString(hipGetLastError()));
    ixtts_gpt_destroy(h);
    return IXTTS_ERR_NOMEM;
  };
  if (hipMalloc(&h->arena, h->arena_bytes) != hipSuccess) return fail("arena");
  if (hipMalloc(&h->stage, h->stage_floats * 4) != hipSuccess) return fail("staging arena");
  const int S = h->slots;
  const size_t kvb = (size_t)h->L * S * D * h->smax * es;
  if (hipMalloc(&h->kc, kvb) != hipSuccess || hipMalloc(&h->vc, kvb) != hipSuccess) return fail("kv cache");
  bool ok = true;
  ok &= hipMalloc(&h->h, (size_t)S * D * 4) == hipSuccess;
  ok &= hipMalloc(&h->h2, (size_t)S * D * 4) == hipSuccess;
  h->hc = h->h;
  ok &= hipMalloc(&h->q, (size_t)S * D * 4) == hipSuccess;
  ok &= hipMalloc(&h->ff, (size_t)S * FF * 4) == hipSuccess;
  ok &= hipMalloc(&h->att, (size_t)S * D * 4) == hipSuccess;
  ok &= hipMalloc(&h->part, (size_t)S * h->H * NSPLIT_MAX * PART_STRIDE * 4) == hipSuccess;
  ok &= hipMalloc(&h->logits, (size_t)S * V * 4) == hipSuccess;
  ok &= hipMalloc(&h->rowbuf, (size_t)D * 4) == hipSuccess;
  ok &= hipMalloc(&h->rx, (size_t)h->smax * D * 4) == hipSuccess;
  ok &= hipMalloc(&h->rxn, (size_t)h->smax * D * 4) == hipSuccess;
  ok &= hipMalloc(&h->rq, (size_t)h->smax * D * 4) == hipSuccess;
  ok &= hipMalloc(&h->ratt, (size_t)h->smax * D * 4) == hipSuccess;
  ok &= hipMalloc(&h->rff, (size_t)h->smax * FF * 4) == hipSuccess;
  ok &= hipMalloc(&h->cur_len, S * 4) == hipSuccess;
  ok &= hipMalloc(&h->gen_count, S * 4) == hipSuccess;
  ok &= hipMalloc(&h->prompt_len, S * 4) == hipSuccess;
  ok &= hipMalloc(&h->valid_from, S * 4) == hipSuccess;
  ok &= hipMalloc(&h->finished, S * 4) == hipSuccess;
  ok &= hipMalloc(&h->forced, S * 4) == hipSuccess;
  ok &= hipMalloc(&h->tokens, (size_t)S * h->smax * 4) == hipSuccess;
  ok &= hipMalloc(&h->seen, (size_t)S * V) == hipSuccess;
  ok &= hipMalloc(&h->d_samp, sizeof(ixtts_sampler_cfg)) == hipSuccess;
  ok &= hipMalloc(&h->beam_scores, MAXB * 4) == hipSuccess;
  ok &= hipMalloc(&h->hyp_score, MAXB * 4) == hipSuccess;
  ok &= hipMalloc(&h->hyp_worst, MAXG * 4) == hipSuccess;
  ok &= hipMalloc(&h->beam_src, MAXB * 4) == hipSuccess;
  ok &= hipMalloc(&h->hyp_len, MAXB * 4) == hipSuccess;
  ok &= hipMalloc(&h->n_hyp, MAXG * 4) == hipSuccess;
  ok &= hipMalloc(&h->beam_done, MAXG * 4) == hipSuccess;
  ok &= hipMalloc(&h->beam_forced_flag, MAXG * 4) == hipSuccess;
  ok &= hipMalloc(&h->beam_forced, (size_t)MAXG * BEAM_FORCED_STRIDE * 4) == hipSuccess;
  ok &= hipMalloc(&h->beam_stream, MAXG * 8) == hipSuccess;
  ok &= hipMalloc(&h->hyp_tok, (size_t)MAXB * h->smax * 4) == hipSuccess;
  ok &= hipMalloc(&h->beam_cand_v, (size_t)MAXB * SAMP_MAXK * 4) == hipSuccess;
  ok &= hipMalloc(&h->beam_cand_i, (size_t)MAXB * SAMP_MAXK * 4) == hipSuccess;
  ok &= hipMalloc(&h->beam_cand_n, (size_t)MAXB * 4) == hipSuccess;
  ok &= hipMalloc(&h->beam_lcp, (size_t)MAXG * BEAM_LCP_STRIDE * 4) == hipSuccess;
  ok &= hipMalloc(&h->probs, (size_t)S * V * 4) == hipSuccess;
  h->scratch_floats = (size_t)FF * D;
  ok &= hipMalloc(&h->scratch, h->scratch_floats * 4) == hipSuccess;
  if (!ok) return fail("state");
  hipMemset(h->cur_len, 0, S * 4);
  hipMemset(h->gen_count, 0, S * 4);
  hipMemset(h->prompt_len, 0, S * 4);
  hipMemset(h->valid_from, 0, S * 4);
  hipMemset(h->finished, 0, S * 4);
  hipMemset(h->forced, 0xff, S * 4);
  hipMemset(h->seen, 0, (size_t)S * V);
  hipMemset(h->logits, 0, (size_t)S * V * 4);
  {  // every beam group starts parked (done): a group below a stepping one that was never begun is a no-op in the beam kernels
    int ones[MAXG];
    for (int g = 0; g < MAXG; ++g) ones[g] = 1;
    hipMemcpy(h->beam_done, ones, sizeof(ones), hipMemcpyHostToDevice);
    hipMemset(h->beam_forced_flag, 0, MAXG * 4);
    hipMemset(h->beam_stream, 0, MAXG * 8);
    hipMemset(h->n_hyp, 0, MAXG * 4);
    hipMemset(h->beam_lcp, 0, (size_t)MAXG * BEAM_LCP_STRIDE * 4);
    hipMemset(h->beam_src, 0, MAXB * 4);
  }
  if (hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) return fail("stream");
  if (const char* e = getenv("IXTTS_ATTN")) h->attn_split = strcmp(e, "legacy") != 0;
  h->wide = c->max_batch > MAXB_REG;
  if (const char* e = getenv("IXTTS_WIDE")) h->wide = h->wide || (strcmp(e, "1") == 0 && c->weight_dtype == IXTTS_DTYPE_BF16);  // A/B: small batches on the MFMA GEMVs
  if (const char* e = getenv("IXTTS_BEAM_REORDER")) h->beam_every_row = strcmp(e, "full") == 0;
  if (const char* e = getenv("IXTTS_PF_KIB")) h->pf_per = atoi(e);  // IXTTS_PF builds: KiB fetched ahead per consumer wave (0: none)
  if (h->wide) h->attn_split = false;  // one workgroup per (head, slot): see forward_layers_wide
  memset(h->host_prompt_len, 0, sizeof(h->host_prompt_len));
  memset(h->host_gen_est, 0, sizeof(h->host_gen_est));
  *out = h;
  return IXTTS_OK;
}

extern "C" int ixtts_gpt_set_tensor(ixtts_gpt* h, const char* name, const float* data, const int64_t* shape, int ndim) {
  IX_ARG(h && name && data && shape, "gpt_set_tensor: null argument");
  auto it = h->tens.find(name);
  if (it == h->tens.end()) {
    set_error("gpt_set_tensor: unknown tensor '%s'", name);
    return IXTTS_ERR_NAME;
  }
  TDesc& t = it->second;
  if (t.kind == T_VEC) {
    IX_ARG(ndim == 1 && shape[0] == t.d0, "gpt_set_tensor: %s expects [%lld]", name, (long long)t.d0);
    IX_HIP(hipMemcpy(h->arena + t.off, data, t.d0 * 4, hipMemcpyHostToDevice));
  } else {
    IX_ARG(ndim == 2 && shape[0] == t.d0 && shape[1] == t.d1, "gpt_set_tensor: %s expects [%lld,%lld]", name, (long long)t.d0, (long long)t.d1);
    const size_t n = (size_t)t.d0 * t.d1;
    if (t.kind == T_EMB) {
      IX_HIP(hipMemcpy(h->arena + t.off, data, n * 4, hipMemcpyHostToDevice));
    } else {
      if (!h->stage) {
        set_error("gpt_set_tensor: %s arrived after finalize", name);
        return IXTTS_ERR_STATE;
      }
      float* dst = h->stage + t.stage_off;
      if (t.kind == T_MAT_T) {  // Conv1D weight [K][N] -> staging Wt[N][K]
        IX_ARG(n <= h->scratch_floats, "gpt_set_tensor: %s larger than the upload buffer", name);
        IX_HIP(hipMemcpy(h->scratch, data, n * 4, hipMemcpyHostToDevice));
        const int K = (int)t.d0, N = (int)t.d1;
        dim3 grid(ceil_div(N, 32), ceil_div(K, 32)), blk(32, 8);
        hipLaunchKernelGGL(pack_transpose_kernel, grid, blk, 0, 0, h->scratch, dst, K, N);
        IX_HIP(hipGetLastError());
        IX_HIP(hipDeviceSynchronize());
      } else {  // nn.Linear weight [N][K]: already row-per-output
        IX_HIP(hipMemcpy(dst, data, n * 4, hipMemcpyHostToDevice));
      }
    }
  }
  t.set = true;
  return IXTTS_OK;
}

template <typename WT>
static int fold_one(ixtts_gpt* h, const char* wname, const float* g, const float* beta, float* bias, size_t dst_off) {
  const TDesc& t = h->tens.at(wname);
  // staging is [N][K]: Conv1D tensors were registered as [K][N], nn.Linear as [N][K]
  const int N = (int)(t.kind == T_MAT_T ? t.d1 : t.d0), K = (int)(t.kind == T_MAT_T ? t.d0 : t.d1);
  hipLaunchKernelGGL(fold_convert_kernel<WT>, dim3(ceil_div(N, 4)), dim3(256), 0, 0, h->stage + t.stage_off, g, beta, bias,
                     reinterpret_cast<WT*>(h->arena + dst_off), N, K);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

template <typename WT>
static int fold_all(ixtts_gpt* h) {
  for (int l = 0; l < h->L; ++l) {
    const LayerOff& o = h->lo[l];
    std::string p = "gpt.h." + std::to_string(l) + ".";
    // LN gain/bias folded into the matrix they feed: W' = W diag(g), b' = b + W beta
    IX_TRY(fold_one<WT>(h, (p + "attn.c_attn.weight").c_str(), A_F32(o.ln1_w), A_F32(o.ln1_b), A_F32(o.bqkv), o.wqkv));
    IX_TRY(fold_one<WT>(h, (p + "attn.c_proj.weight").c_str(), nullptr, nullptr, nullptr, o.wo));
    IX_TRY(fold_one<WT>(h, (p + "mlp.c_fc.weight").c_str(), A_F32(o.ln2_w), A_F32(o.ln2_b), A_F32(o.bfc), o.wfc));
    IX_TRY(fold_one<WT>(h, (p + "mlp.c_proj.weight").c_str(), nullptr, nullptr, nullptr, o.wpr));
  }
  // final_norm folds into mel_head; ln_f stays explicit (a LayerNorm sits between them)
  IX_TRY(fold_one<WT>(h, "mel_head.weight", A_F32(h->fn_w), A_F32(h->fn_b), A_F32(h->bhead), h->whead));
  IX_HIP(hipDeviceSynchronize());
  return IXTTS_OK;
}

// The launch fused on the XCD-local hand-off, opt-in: IXTTS_MLP=fused (mlp_fused_kernel: LN2 + c_fc + gelu + c_proj, 315 MB
// for the per-XCD copy of c_proj; measured -1.7 % per step at B=2 and nothing at B=1, profiles/r01_spikes.md).  It needs
// bf16 weights at model_dim 1280 and workgroups dealt round-robin over the 8 XCDs -- probed here with HW_REG_XCC_ID;
// anything else keeps the split kernels.  Called once the arena holds the final weights.
static int derive_fused_mlp(ixtts_gpt* h) {
  h->mlp_fused = false;
  static const bool dbg = getenv("IXTTS_DEBUG") != nullptr;
  if (h->esize != 2 || h->D != MLP_D || h->FF != MLP_FF || h->L < 2 || h->wide) return IXTTS_OK;
  const char* m_mlp = getenv("IXTTS_MLP");
  if (!m_mlp || strcmp(m_mlp, "fused")) return IXTTS_OK;
  IX_HIP(hipDeviceSynchronize());  // (a broadcast into the arena may still be in flight on another stream)
  const size_t ctr_uints = (size_t)h->L * MLP_CTR_STRIDE + 4096;  // per-layer counters, then the time-out mark (+ developer log)
  if (!h->mlp_ctr && hipMalloc(&h->mlp_ctr, ctr_uints * 4) != hipSuccess) {
    set_error("gpt: allocation failed (hand-off counters): %s", hipGetErrorString(hipGetLastError()));
    return IXTTS_ERR_NOMEM;
  }
  IX_HIP(hipMemset(h->mlp_ctr, 0, ctr_uints * 4));
  // XCD placement probe (a few launches of the decode grid): workgroups must be dealt round-robin over the 8 XCDs, from any start
  unsigned* map = h->mlp_ctr + (size_t)h->L * MLP_CTR_STRIDE + 1024;  // (scratch inside the spare region; cleared again below)
  bool dealt = true;
  for (int i = 0; i < 8 && dealt; ++i) {
    hipLaunchKernelGGL(mlp_xcc_probe_kernel, dim3(256), dim3(64 * MLP_WAVES), 0, 0, map);
    if (i & 1) hipLaunchKernelGGL(mlp_xcc_probe_kernel, dim3(3), dim3(64), 0, 0, map + 256);  // shift the dealer's start
    unsigned m[256];
    IX_HIP(hipMemcpy(m, map, sizeof(m), hipMemcpyDeviceToHost));
    for (int w = 0; w < 256; ++w) dealt = dealt && m[w] < 8u && m[w] == ((m[0] + (unsigned)w) & 7u);
    if (dbg) fprintf(stderr, "[ixtts] fused launches: probe %d: workgroup 0 on XCD %u, round-robin %s\n", i, m[0], dealt ? "yes" : "NO");
  }
  IX_HIP(hipMemset(h->mlp_ctr, 0, ctr_uints * 4));
  if (!dealt) return IXTTS_OK;  // keep the split kernels
  const size_t per_layer = (size_t)MLP_D * MLP_FF * sizeof(bf16);
  if (!h->wprx) {
    bool ok = hipMalloc(&h->wprx, per_layer * (h->L - 1)) == hipSuccess;
    ok = ok && hipMalloc(&h->mlp_part, (size_t)MLP_XCDS * h->slots * MLP_D * 4) == hipSuccess;
    if (!ok) {
      set_error("gpt: allocation failed (fused MLP tables): %s", hipGetErrorString(hipGetLastError()));
      return IXTTS_ERR_NOMEM;
    }
  }
  IX_HIP(hipMemset(h->mlp_part, 0, (size_t)MLP_XCDS * h->slots * MLP_D * 4));
  constexpr size_t G = (size_t)MLP_XCDS * MLP_D * MLP_SLICE / 8;
  for (int l = 0; l + 1 < h->L; ++l)
    hipLaunchKernelGGL(mlp_repack_pr_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<const bf16*>(A_PTR(h->lo[l].wpr)),
                       reinterpret_cast<bf16*>(h->wprx) + (size_t)l * MLP_D * MLP_FF);
  IX_HIP(hipDeviceSynchronize());
  IX_HIP(hipGetLastError());
  h->mlp_fused = true;
  if (dbg) fprintf(stderr, "[ixtts] fused MLP: on\n");
  return IXTTS_OK;
}

extern "C" int ixtts_gpt_finalize(ixtts_gpt* h) {
  IX_ARG(h, "gpt_finalize: null handle");
  if (h->finalized) return IXTTS_OK;
  for (auto& kv : h->tens)
    if (!kv.second.set) {
      set_error("gpt_finalize: tensor '%s' was not supplied", kv.first.c_str());
      return IXTTS_ERR_STATE;
    }
  if (h->esize == 4) IX_TRY(fold_all<float>(h));
  else IX_TRY(fold_all<bf16>(h));
  hipFree(h->stage);
  h->stage = nullptr;
  h->finalized = true;
  return derive_fused_mlp(h);
}

extern "C" int ixtts_gpt_arena(ixtts_gpt* h, void** ptr, size_t* bytes) {
  IX_ARG(h && ptr && bytes, "gpt_arena: null argument");
  *ptr = h->arena;
  *bytes = h->arena_bytes;
  return IXTTS_OK;
}

extern "C" int ixtts_gpt_adopt_arena(ixtts_gpt* h) {
  IX_ARG(h, "gpt_adopt_arena: null handle");
  for (auto& kv : h->tens) kv.second.set = true;
  if (h->stage) {
    hipFree(h->stage);
    h->stage = nullptr;
  }
  h->finalized = true;
  return derive_fused_mlp(h);
}

// A second engine over the SAME weights (e.g. the wide beam-group engine beside the register engine of one worker): `h`
// drops its own arena and reads `owner`'s, which must be finalized, of the same shape and type, and outlive `h`.
extern "C" int ixtts_gpt_share_arena(ixtts_gpt* h, ixtts_gpt* owner) {
  IX_ARG(h && owner && h != owner, "gpt_share_arena: bad handles");
  IX_ARG(owner->finalized && !owner->arena_borrowed, "gpt_share_arena: the owner must hold finalized weights of its own");
  IX_ARG(h->D == owner->D && h->L == owner->L && h->H == owner->H && h->V == owner->V && h->esize == owner->esize &&
             h->cfg.n_mel_pos == owner->cfg.n_mel_pos && h->arena_bytes == owner->arena_bytes,
         "gpt_share_arena: the two engines differ in shape or weight type");
  IX_HIP(hipDeviceSynchronize());
  if (h->arena && !h->arena_borrowed) hipFree(h->arena);
  h->arena = owner->arena;
  h->arena_borrowed = true;
  for (auto& kv : h->tens) kv.second.set = true;
  if (h->stage) {
    hipFree(h->stage);
    h->stage = nullptr;
  }
  h->finalized = true;
  return derive_fused_mlp(h);
}

#define NEED_READY(h, who)                                       \
  do {                                                           \
    IX_ARG(h, who ": null handle");                              \
    if (!(h)->finalized) {                                       \
      set_error(who ": weights not finalized");                  \
      return IXTTS_ERR_STATE;                                    \
    }                                                            \
  } while (0)

// ------------------------------------------------------------------------------------ prefill
namespace ixtts {
__global__ void add_vec_kernel(float* dst, const float* a, const float* b, int D) {
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < D; i += blockDim.x * gridDim.x) dst[i] = a[i] + b[i];
}
}  // namespace ixtts
extern "C" int ixtts_gpt_prefill(ixtts_gpt* h, int b, const float* embeds, int n_rows, int n_left_pad, void* stream) {
  NEED_READY(h, "gpt_prefill");
  IX_ARG(b >= 0 && b < h->cfg.max_batch, "gpt_prefill: slot %d out of range", b);
  IX_ARG(embeds && n_rows >= 1 && n_left_pad >= 0 && n_left_pad < n_rows, "gpt_prefill: bad rows (%d, pad %d)", n_rows, n_left_pad);
  IX_ARG(n_rows + 2 < h->smax, "gpt_prefill: prompt of %d rows exceeds max_seq %d", n_rows + 1, h->smax);
  hipStream_t st = (hipStream_t)stream;
  const int D = h->D, V = h->V;
  // slot state: history = fake ids [1]*(P-1) + [start]  (model_v2.py:652-661)
  IX_HIP(hipMemsetAsync(h->seen + (size_t)b * V, 0, V, st));
  IX_HIP(hipMemsetAsync(h->seen + (size_t)b * V + 1, 1, 1, st));
  IX_HIP(hipMemsetAsync(h->seen + (size_t)b * V + h->cfg.start_mel_token, 1, 1, st));
  IX_HIP(hipMemsetAsync(h->gen_count + b, 0, 4, st));
  IX_HIP(hipMemsetAsync(h->finished + b, 0, 4, st));
  IX_HIP(hipMemsetAsync(h->forced + b, 0xff, 4, st));
  const int P = n_rows + 1;
  IX_HIP(hipMemcpyAsync(h->prompt_len + b, &P, 4, hipMemcpyHostToDevice, st));
  IX_HIP(hipMemcpyAsync(h->valid_from + b, &n_left_pad, 4, hipMemcpyHostToDevice, st));
  h->host_prompt_len[b] = P;
  h->host_gen_est[b] = 0;
  // Batched causal pass over the un-padded rows (left-pad rows are never attended to --
  // valid_from masks them as keys -- so they are skipped outright).
  const int T = P - n_left_pad;
  IX_HIP(hipMemcpyAsync(h->rx, embeds + (size_t)n_left_pad * D, (size_t)(n_rows - n_left_pad) * D * 4, hipMemcpyDeviceToDevice, st));
  // start_mel_token row: mel_embedding[start] + mel_pos_embedding[0]   (model_v2.py:146-148)
  hipLaunchKernelGGL(add_vec_kernel, dim3(4), dim3(256), 0, st, h->rx + (size_t)(T - 1) * D,
                     (const float*)(A_F32(h->mel_emb) + (size_t)h->cfg.start_mel_token * D), (const float*)A_F32(h->mel_pos), D);
  IX_TRY(forward_rows(h, b, T, n_left_pad, n_left_pad, st));
  IX_HIP(hipMemcpyAsync(h->h + (size_t)b * D, h->rx + (size_t)(T - 1) * D, (size_t)D * 4, hipMemcpyDeviceToDevice, st));
  h->hc = h->h;
  IX_TRY(do_head(h, 1, b, nullptr, st));
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

// ------------------------------------------------------------------------------------ decode
// One graph = `reps` consecutive decode steps (sampler -> 24 layers -> head, 122 kernels each): replaying an
// 8-step graph amortises the ~10-16 us host cost of a graph launch over 8 tokens.
static int build_step_graph(ixtts_gpt* h, int B, int reps, int bucket, hipGraphExec_t* out, bool beam = false) {
  hipGraph_t g;
  hipStream_t cs = h->cap_stream;
  h->attn_bucket = bucket;
  IX_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
  int rc = IXTTS_OK;
  for (int r = 0; r < reps && rc == IXTTS_OK; ++r) {
    if (beam) launch_beam_step(h, make_sampler_state(h), cs);
    else launch_sampler(h, B, cs);
    rc = do_forward_layers(h, B, 0, cs);
    if (rc == IXTTS_OK) rc = do_head(h, B, 0, nullptr, cs);
  }
  hipError_t e = hipStreamEndCapture(cs, &g);
  if (rc != IXTTS_OK) return rc;
  IX_HIP(e);
  IX_HIP(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
  IX_HIP(hipGraphDestroy(g));
  return IXTTS_OK;
}

// smallest attention bucket that holds `ctx` keys (NBKT: the any-length kernel)
static int pick_bucket(const ixtts_gpt* h, int ctx) {
  if (!h->attn_split) return NBKT;
  for (int k = 0; k < NBKT; ++k)
    if (ctx <= attn_cover(k)) return k;
  return NBKT;
}

extern "C" int ixtts_gpt_decode(ixtts_gpt* h, int n_active, int n_steps, const ixtts_sampler_cfg* sc, void* stream) {
  NEED_READY(h, "gpt_decode");
  IX_ARG(sc, "gpt_decode: null sampler cfg");
  IX_ARG(n_active >= 1 && n_active <= h->cfg.max_batch, "gpt_decode: n_active %d", n_active);
  IX_ARG(n_steps >= 0, "gpt_decode: n_steps %d", n_steps);
  if (sc->do_sample) {
    IX_ARG(sc->temperature > 0.f && sc->top_p > 0.f, "gpt_decode: temperature and top_p must be positive");
  }
  IX_ARG(h->V <= 1024 * SAMP_PT, "gpt_decode: vocabulary %d exceeds the sampler tile", h->V);
  for (int b = 0; b < n_active; ++b) {
    IX_ARG(h->host_prompt_len[b] > 0, "gpt_decode: slot %d has no prefilled prompt", b);
    IX_ARG(h->host_prompt_len[b] + h->host_gen_est[b] + n_steps < h->smax, "gpt_decode: slot %d would overflow max_seq %d", b, h->smax);
  }
  hipStream_t st = (hipStream_t)stream;
  h->samp_host = *sc;
  IX_HIP(hipMemcpyAsync(h->d_samp, &h->samp_host, sizeof(ixtts_sampler_cfg), hipMemcpyHostToDevice, st));
  int ctx0 = 0;  // keys the longest sequence holds before this call (the host counts the steps it issues)
  for (int b = 0; b < n_active; ++b) ctx0 = std::max(ctx0, h->host_prompt_len[b] + h->host_gen_est[b]);
  for (int done = 0; done < n_steps;) {
    const int reps = n_steps - done >= STEPS_PER_GRAPH ? STEPS_PER_GRAPH : 1;
    const int bkt = pick_bucket(h, ctx0 + done + reps + 1);
    hipGraphExec_t* slot = reps > 1 ? &h->multi_exec[n_active][bkt] : &h->step_exec[n_active][bkt];
    if (!*slot) IX_TRY(build_step_graph(h, n_active, reps, bkt, slot));
    IX_HIP(hipGraphLaunch(*slot, st));
    done += reps;
  }
  for (int b = 0; b < n_active; ++b) h->host_gen_est[b] += n_steps;
  return IXTTS_OK;
}

// ------------------------------------------------------------------------------------ beam-sample
// Group g = the beams of ONE prompt, in slots g*NB .. g*NB+NB-1.  Engines of up to 4 slots hold one group (register GEMVs);
// wide engines (5..16 slots, bf16) hold floor(max_batch / NB) groups that step together: the text segments of a request, or
// of several requests, each with its own scorer state (the reference decodes them one after another, infer_v2.py:616).
static int beam_begin_impl(ixtts_gpt* h, int g, int num_beams, unsigned long long stream_id, hipStream_t st) {
  IX_ARG(num_beams >= 2 && num_beams <= BEAM_MAX, "gpt_beam_begin: num_beams %d (2..%d)", num_beams, BEAM_MAX);
  IX_ARG(g >= 0 && g < MAXG && (g + 1) * num_beams <= h->cfg.max_batch, "gpt_beam_begin: group %d of %d beams needs max_batch >= %d (have %d)", g, num_beams,
         (g + 1) * num_beams, h->cfg.max_batch);
  for (int o = 0; o < MAXG; ++o)
    if (o != g && h->group_live[o]) IX_ARG(h->num_beams == num_beams, "gpt_beam_begin: live groups hold %d beams each, group %d asks for %d", h->num_beams, g, num_beams);
  const int sb = g * num_beams;
  IX_ARG(h->host_prompt_len[sb] > 0, "gpt_beam_begin: slot %d (first of group %d) has no prefilled prompt", sb, g);
  const int D = h->D, V = h->V;
  const size_t slot_bytes = (size_t)D * h->smax * h->esize;
  const size_t layer_bytes = (size_t)h->slots * slot_bytes;
  // input_ids.repeat_interleave(num_beams): every beam starts as a copy of the prefilled sequence
  for (int b = sb + 1; b < sb + num_beams; ++b) {
    for (int l = 0; l < h->L; ++l) {
      IX_HIP(hipMemcpyAsync((char*)h->kc + l * layer_bytes + b * slot_bytes, (char*)h->kc + l * layer_bytes + sb * slot_bytes, slot_bytes, hipMemcpyDeviceToDevice, st));
      IX_HIP(hipMemcpyAsync((char*)h->vc + l * layer_bytes + b * slot_bytes, (char*)h->vc + l * layer_bytes + sb * slot_bytes, slot_bytes, hipMemcpyDeviceToDevice, st));
    }
    IX_HIP(hipMemcpyAsync(h->logits + (size_t)b * V, h->logits + (size_t)sb * V, (size_t)V * 4, hipMemcpyDeviceToDevice, st));
    IX_HIP(hipMemcpyAsync(h->seen + (size_t)b * V, h->seen + (size_t)sb * V, V, hipMemcpyDeviceToDevice, st));
    IX_HIP(hipMemcpyAsync(h->prompt_len + b, h->prompt_len + sb, 4, hipMemcpyDeviceToDevice, st));
    IX_HIP(hipMemcpyAsync(h->valid_from + b, h->valid_from + sb, 4, hipMemcpyDeviceToDevice, st));
    IX_HIP(hipMemcpyAsync(h->gen_count + b, h->gen_count + sb, 4, hipMemcpyDeviceToDevice, st));
    IX_HIP(hipMemcpyAsync(h->finished + b, h->finished + sb, 4, hipMemcpyDeviceToDevice, st));
    h->host_prompt_len[b] = h->host_prompt_len[sb];
    h->host_gen_est[b] = h->host_gen_est[sb];
  }
  // beam_scores = [0, -1e9, ...] so only beam 0's tokens can be drawn at the first step (generation_utils.py:3406-3410)
  float bs[BEAM_MAX];
  for (int b = 0; b < BEAM_MAX; ++b) bs[b] = b == 0 ? 0.f : -1e9f;
  IX_HIP(hipMemcpyAsync(h->beam_scores + sb, bs, num_beams * 4, hipMemcpyHostToDevice, st));
  const float worst = 1e9f;
  IX_HIP(hipMemcpyAsync(h->hyp_worst + g, &worst, 4, hipMemcpyHostToDevice, st));
  IX_HIP(hipMemcpyAsync(h->beam_stream + g, &stream_id, 8, hipMemcpyHostToDevice, st));
  IX_HIP(hipMemsetAsync(h->n_hyp + g, 0, 4, st));
  IX_HIP(hipMemsetAsync(h->beam_done + g, 0, 4, st));
  IX_HIP(hipMemsetAsync(h->beam_forced_flag + g, 0, 4, st));
  IX_HIP(hipMemsetAsync(h->beam_lcp + (size_t)g * BEAM_LCP_STRIDE, 0, (size_t)BEAM_LCP_STRIDE * 4, st));  // no generated rows yet
  IX_HIP(hipStreamSynchronize(st));  // bs / worst / stream_id are stack variables
  h->num_beams = num_beams;
  h->group_live[g] = h->group_begun[g] = true;
  return IXTTS_OK;
}

extern "C" int ixtts_gpt_beam_begin(ixtts_gpt* h, int num_beams, void* stream) {
  NEED_READY(h, "gpt_beam_begin");
  return beam_begin_impl(h, 0, num_beams, 0ull, (hipStream_t)stream);
}

extern "C" int ixtts_gpt_beam_begin_group(ixtts_gpt* h, int group, int num_beams, uint64_t rng_stream, void* stream) {
  NEED_READY(h, "gpt_beam_begin_group");
  return beam_begin_impl(h, group, num_beams, (unsigned long long)rng_stream, (hipStream_t)stream);
}

// A group that has no prompt to work on while others step: its kernels are no-ops (done flag), its slots keep their last
// K/V position, and its context no longer counts for the overflow check.
extern "C" int ixtts_gpt_beam_park_group(ixtts_gpt* h, int group, void* stream) {
  NEED_READY(h, "gpt_beam_park_group");
  IX_ARG(group >= 0 && group < MAXG, "gpt_beam_park_group: group %d", group);
  hipStream_t st = (hipStream_t)stream;
  static const int one = 1;
  IX_HIP(hipMemcpyAsync(h->beam_done + group, &one, 4, hipMemcpyHostToDevice, st));
  h->group_live[group] = false;
  return IXTTS_OK;
}

static int beam_force_impl(ixtts_gpt* h, int g, const int32_t* picks, int n, hipStream_t st) {
  IX_ARG(h->num_beams >= 2 && g >= 0 && g < MAXG && h->group_live[g] && picks && n == 2 * h->num_beams, "gpt_beam_force: need 2*num_beams picks for a begun group");
  const int one = 1;
  IX_HIP(hipMemcpyAsync(h->beam_forced + (size_t)g * BEAM_FORCED_STRIDE, picks, n * 4, hipMemcpyHostToDevice, st));
  IX_HIP(hipMemcpyAsync(h->beam_forced_flag + g, &one, 4, hipMemcpyHostToDevice, st));
  IX_HIP(hipStreamSynchronize(st));
  return IXTTS_OK;
}

extern "C" int ixtts_gpt_beam_force(ixtts_gpt* h, const int32_t* picks, int n, void* stream) {
  NEED_READY(h, "gpt_beam_force");
  return beam_force_impl(h, 0, picks, n, (hipStream_t)stream);
}

extern "C" int ixtts_gpt_beam_force_group(ixtts_gpt* h, int group, const int32_t* picks, int n, void* stream) {
  NEED_READY(h, "gpt_beam_force_group");
  return beam_force_impl(h, group, picks, n, (hipStream_t)stream);
}

static int beam_decode_impl(ixtts_gpt* h, int n_groups, int n_steps, const ixtts_sampler_cfg* sc, hipStream_t st) {
  IX_ARG(sc && n_steps >= 0, "gpt_beam_decode: bad argument");
  IX_ARG(h->num_beams >= 2, "gpt_beam_decode: call gpt_beam_begin first");
  // do_sample == 0 is beam search proper: the warpers do not run (generation_utils.py:1020), their settings are not looked at
  IX_ARG(!sc->do_sample || (sc->top_k >= 1 && sc->top_k <= SAMP_MAXK && sc->temperature > 0.f && sc->top_p > 0.f),
         "gpt_beam_decode: 1 <= top_k <= %d, positive temperature/top_p", SAMP_MAXK);
  IX_ARG(h->V <= 1024 * SAMP_PT, "gpt_beam_decode: vocabulary %d exceeds the sampler tile", h->V);
  const int nb = h->num_beams;
  IX_ARG(n_groups >= 1 && n_groups <= MAXG && n_groups * nb <= h->cfg.max_batch, "gpt_beam_decode: %d groups of %d beams exceed max_batch %d", n_groups, nb, h->cfg.max_batch);
  IX_ARG(n_groups == 1 || h->wide, "gpt_beam_decode: several groups step together on the wide engines only (max_batch > %d, bf16)", MAXB_REG);
  int ctx0 = 0, live = 0;
  for (int g = 0; g < n_groups; ++g) {
    if (!h->group_live[g]) continue;  // parked (or never begun: created parked)
    ++live;
    const int c = h->host_prompt_len[g * nb] + h->host_gen_est[g * nb];
    IX_ARG(c + n_steps < h->smax, "gpt_beam_decode: group %d would overflow max_seq %d", g, h->smax);
    ctx0 = std::max(ctx0, c);
  }
  IX_ARG(live > 0, "gpt_beam_decode: none of the %d groups is live (gpt_beam_begin_group first)", n_groups);
  if (h->beam_exec_nb != nb) {
    for (int g = 0; g <= MAXG; ++g)
      for (int k = 0; k <= NBKT; ++k) {
        if (h->beam_exec[g][k]) hipGraphExecDestroy(h->beam_exec[g][k]);
        if (h->beam_multi_exec[g][k]) hipGraphExecDestroy(h->beam_multi_exec[g][k]);
        h->beam_exec[g][k] = h->beam_multi_exec[g][k] = nullptr;
      }
    h->beam_exec_nb = nb;
  }
  h->samp_host = *sc;
  IX_HIP(hipMemcpyAsync(h->d_samp, &h->samp_host, sizeof(ixtts_sampler_cfg), hipMemcpyHostToDevice, st));
  h->beam_groups = n_groups;
  for (int done = 0; done < n_steps;) {
    const int reps = n_steps - done >= STEPS_PER_GRAPH ? STEPS_PER_GRAPH : 1;
    const int bkt = pick_bucket(h, ctx0 + done + reps + 1);
    hipGraphExec_t* slot = reps > 1 ? &h->beam_multi_exec[n_groups][bkt] : &h->beam_exec[n_groups][bkt];
    if (!*slot) IX_TRY(build_step_graph(h, n_groups * nb, reps, bkt, slot, true));
    IX_HIP(hipGraphLaunch(*slot, st));
    done += reps;
  }
  for (int g = 0; g < n_groups; ++g)
    if (h->group_live[g])
      for (int b = g * nb; b < (g + 1) * nb; ++b) h->host_gen_est[b] += n_steps;
  return IXTTS_OK;
}

extern "C" int ixtts_gpt_beam_decode(ixtts_gpt* h, int n_steps, const ixtts_sampler_cfg* sc, void* stream) {
  NEED_READY(h, "gpt_beam_decode");
  return beam_decode_impl(h, 1, n_steps, sc, (hipStream_t)stream);
}

extern "C" int ixtts_gpt_beam_decode_groups(ixtts_gpt* h, int n_groups, int n_steps, const ixtts_sampler_cfg* sc, void* stream) {
  NEED_READY(h, "gpt_beam_decode_groups");
  return beam_decode_impl(h, n_groups, n_steps, sc, (hipStream_t)stream);
}

// BeamSearchScorer.finalize (transformers_beam_search.py:320-417), num_return_sequences = 1, on the host.
static int beam_read_impl(ixtts_gpt* h, int g, int max_new, int32_t* ids, int cap, int* n_ids, int* done, float* score,
                          float* beam_scores_out, int32_t* last_tokens_out, int32_t* src_out, hipStream_t st) {
  IX_ARG(h->num_beams >= 2 && ids && n_ids && done && cap >= 0 && max_new >= 0, "gpt_beam_read: bad argument");
  IX_ARG(g >= 0 && g < MAXG && h->group_begun[g], "gpt_beam_read: group %d was never begun", g);
  IX_HIP(hipStreamSynchronize(st));
  const int nb = h->num_beams;
  const int sb = g * nb;
  int gc = 0, dn = 0, nh = 0;
  float bs[MAXB], hs[MAXB];
  int hl[MAXB], src[MAXB];
  IX_HIP(hipMemcpy(&gc, h->gen_count + sb, 4, hipMemcpyDeviceToHost));
  IX_HIP(hipMemcpy(&dn, h->beam_done + g, 4, hipMemcpyDeviceToHost));
  IX_HIP(hipMemcpy(&nh, h->n_hyp + g, 4, hipMemcpyDeviceToHost));
  IX_HIP(hipMemcpy(bs, h->beam_scores + sb, nb * 4, hipMemcpyDeviceToHost));
  IX_HIP(hipMemcpy(hs, h->hyp_score + sb, nb * 4, hipMemcpyDeviceToHost));
  IX_HIP(hipMemcpy(hl, h->hyp_len + sb, nb * 4, hipMemcpyDeviceToHost));
  IX_HIP(hipMemcpy(src, h->beam_src + sb, nb * 4, hipMemcpyDeviceToHost));
  gc = std::min(gc, h->smax);
  nh = std::max(0, std::min(nh, nb));
  std::vector<std::vector<int32_t>> open(nb), hyp(nh);
  for (int b = 0; b < nb; ++b) {
    open[b].resize(gc);
    if (gc) IX_HIP(hipMemcpy(open[b].data(), h->tokens + (size_t)(sb + b) * h->smax, (size_t)gc * 4, hipMemcpyDeviceToHost));
  }
  for (int i = 0; i < nh; ++i) {
    hyp[i].resize(hl[i]);
    if (hl[i]) IX_HIP(hipMemcpy(hyp[i].data(), h->hyp_tok + (size_t)(sb + i) * h->smax, (size_t)hl[i] * 4, hipMemcpyDeviceToHost));
  }
  if (beam_scores_out) memcpy(beam_scores_out, bs, nb * 4);
  if (src_out) memcpy(src_out, src, nb * 4);
  if (last_tokens_out)
    for (int b = 0; b < nb; ++b) last_tokens_out[b] = gc ? open[b][gc - 1] : -1;
  // candidates: finished hypotheses, plus (if not done) the open beams through BeamHypotheses.add
  std::vector<std::pair<float, std::vector<int32_t>>> beams;
  for (int i = 0; i < nh; ++i) beams.emplace_back(hs[i], hyp[i]);
  float worst = 1e9f;
  for (auto& bsc : beams) worst = std::min(worst, bsc.first);
  const float lp = h->samp_host.length_penalty;  // of the last beam_decode call
  if (!dn) {
    for (int b = 0; b < nb; ++b) {
      // finalize: beam_hyp.add(final_tokens, final_score, generated_len = tokens generated)  (transformers_beam_search.py:360-371)
      const float sc = (lp != 0.f && gc > 0) ? bs[b] / powf((float)gc, lp) : bs[b];
      if ((int)beams.size() < nb || sc > worst) {
        beams.emplace_back(sc, open[b]);
        if ((int)beams.size() > nb) {
          int wi = 0;
          for (int i = 1; i < (int)beams.size(); ++i)
            if (beams[i].first < beams[wi].first) wi = i;
          beams.erase(beams.begin() + wi);
          worst = 1e9f;
          for (auto& bb : beams) worst = std::min(worst, bb.first);
        } else {
          worst = std::min(sc, worst);
        }
      }
    }
  }
  IX_ARG(!beams.empty(), "gpt_beam_read: no hypothesis available");
  // sorted(key=score).pop(): the LAST of the maximal scores in insertion order
  int bi = 0;
  for (int i = 1; i < (int)beams.size(); ++i)
    if (beams[i].first >= beams[bi].first) bi = i;
  std::vector<int32_t> seq = beams[bi].second;
  if ((int)seq.size() < max_new) seq.push_back(h->cfg.stop_mel_token);
  const int n = std::min((int)seq.size(), cap);
  for (int i = 0; i < n; ++i) ids[i] = seq[i];
  *n_ids = n;
  *done = dn;
  if (score) *score = beams[bi].first;
  return IXTTS_OK;
}

extern "C" int ixtts_gpt_beam_read(ixtts_gpt* h, int max_new, int32_t* ids, int cap, int* n_ids, int* done, float* score,
                                   float* beam_scores_out, int32_t* last_tokens_out, int32_t* src_out, void* stream) {
  NEED_READY(h, "gpt_beam_read");
  return beam_read_impl(h, 0, max_new, ids, cap, n_ids, done, score, beam_scores_out, last_tokens_out, src_out, (hipStream_t)stream);
}

extern "C" int ixtts_gpt_beam_read_group(ixtts_gpt* h, int group, int max_new, int32_t* ids, int cap, int* n_ids, int* done, float* score,
                                         float* beam_scores_out, int32_t* last_tokens_out, int32_t* src_out, void* stream) {
  NEED_READY(h, "gpt_beam_read_group");
  return beam_read_impl(h, group, max_new, ids, cap, n_ids, done, score, beam_scores_out, last_tokens_out, src_out, (hipStream_t)stream);
}

// The fused MLP kernel never waits forever for its XCD's workgroups: after ~13 ms it leaves a mark and goes on with what it
// has.  Results of such a run are wrong; every read of results checks the mark and fails loudly.
static int check_mlp_handoff(ixtts_gpt* h) {
  if (!h->mlp_fused) return IXTTS_OK;
  unsigned m[1 + MLP_XCDS];
  IX_HIP(hipMemcpy(m, h->mlp_ctr + (size_t)h->L * MLP_CTR_STRIDE, sizeof(m), hipMemcpyDeviceToHost));
#ifdef IXTTS_MLP_LOG
  {  // developer timeline of the layer-10 kernel's last run: per workgroup entry / ff done / arrived / released / end
    static unsigned long long lg[256 * 6];
    IX_HIP(hipMemcpy(lg, h->mlp_ctr + (size_t)(h->L + 1) * MLP_CTR_STRIDE, sizeof(lg), hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < 256; ++i)
      if (lg[6 * i + 1] && lg[6 * i + 1] < t0) t0 = lg[6 * i + 1];
    double s[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 256; ++i)
      for (int k = 0; k < 5; ++k) {
        const double v = (double)(lg[6 * i + 1 + k] - t0) * 0.01;
        s[k] += v / 256;
        mx[k] = v > mx[k] ? v : mx[k];
      }
    fprintf(stderr, "[ixtts] fused MLP timeline us (mean / max over workgroups): entry %.2f/%.2f  ff done %.2f/%.2f  arrived %.2f/%.2f  released %.2f/%.2f  end %.2f/%.2f\n",
            s[0], mx[0], s[1], mx[1], s[2], mx[2], s[3], mx[3], s[4], mx[4]);
  }
#endif
  if (m[0] == 0) return IXTTS_OK;
  set_error("gpt: a fused kernel's hand-off inside an XCD timed out (%u workgroups; arrivals seen per XCD: %u %u %u %u %u %u %u %u of 32); unset IXTTS_MLP",
            m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8]);
  return IXTTS_ERR_STATE;
}

extern "C" int ixtts_gpt_read(ixtts_gpt* h, int b, int32_t* ids, int cap, int* n_ids, int* finished, void* stream) {
  NEED_READY(h, "gpt_read");
  IX_ARG(b >= 0 && b < h->cfg.max_batch && ids && n_ids && finished && cap >= 0, "gpt_read: bad argument");
  hipStream_t st = (hipStream_t)stream;
  IX_HIP(hipStreamSynchronize(st));
  IX_TRY(check_mlp_handoff(h));
  int gc = 0, fin = 0;
  IX_HIP(hipMemcpy(&gc, h->gen_count + b, 4, hipMemcpyDeviceToHost));
  IX_HIP(hipMemcpy(&fin, h->finished + b, 4, hipMemcpyDeviceToHost));
  int n = std::min(gc, std::min(cap, h->smax));
  if (n > 0) IX_HIP(hipMemcpy(ids, h->tokens + (size_t)b * h->smax, (size_t)n * 4, hipMemcpyDeviceToHost));
  int len = n;
  for (int i = 0; i < n; ++i)
    if (ids[i] == h->cfg.stop_mel_token) {
      len = i + 1;
      break;
    }
  *n_ids = len;
  *finished = fin;
  return IXTTS_OK;
}

extern "C" int ixtts_gpt_read_logits(ixtts_gpt* h, int b, float* out, void* stream) {
  NEED_READY(h, "gpt_read_logits");
  IX_ARG(b >= 0 && b < h->cfg.max_batch && out, "gpt_read_logits: bad argument");
  IX_HIP(hipStreamSynchronize((hipStream_t)stream));
  IX_HIP(hipMemcpy(out, h->logits + (size_t)b * h->V, (size_t)h->V * 4, hipMemcpyDeviceToHost));
  return IXTTS_OK;
}

extern "C" int ixtts_gpt_read_probs(ixtts_gpt* h, int b, float* out, void* stream) {
  NEED_READY(h, "gpt_read_probs");
  IX_ARG(b >= 0 && b < h->cfg.max_batch && out, "gpt_read_probs: bad argument");
  IX_HIP(hipStreamSynchronize((hipStream_t)stream));
  IX_HIP(hipMemcpy(out, h->probs + (size_t)b * h->V, (size_t)h->V * 4, hipMemcpyDeviceToHost));
  return IXTTS_OK;
}

extern "C" int ixtts_gpt_force_next(ixtts_gpt* h, int b, int32_t token, void* stream) {
  NEED_READY(h, "gpt_force_next");
  IX_ARG(b >= 0 && b < h->cfg.max_batch && token >= 0 && token < h->V, "gpt_force_next: bad argument");
  hipStream_t st = (hipStream_t)stream;
  IX_HIP(hipMemcpyAsync(h->forced + b, &token, 4, hipMemcpyHostToDevice, st));
  IX_HIP(hipStreamSynchronize(st));
  return IXTTS_OK;
}

// ------------------------------------------------------------------------------------ latent pass
extern "C" int ixtts_gpt_latent(ixtts_gpt* h, const float* prefix, int n_prefix, const int32_t* codes, int n, float* latent,
                                void* stream) {
  NEED_READY(h, "gpt_latent");
  IX_ARG(prefix && n_prefix >= 1 && n >= 0 && (codes || n == 0) && (latent || n == 0), "gpt_latent: bad argument");
  IX_ARG(n + 2 <= h->cfg.n_mel_pos, "gpt_latent: %d codes exceed the mel position table", n);
  IX_ARG(n_prefix + n + 2 < h->smax, "gpt_latent: sequence of %d rows exceeds max_seq %d", n_prefix + n + 2, h->smax);
  if (n == 0) return IXTTS_OK;
  hipStream_t st = (hipStream_t)stream;
  const int D = h->D;
  const int slot = h->slots - 1;  // scratch sequence
  // Only the first n mel rows are returned ([:-2], model_v2.py:596) and the pass is causal,
  // so the two trailing rows (last code, stop) need not be computed at all.
  const int rows = n_prefix + n;
  IX_HIP(hipMemcpyAsync(h->rx, prefix, (size_t)n_prefix * D * 4, hipMemcpyDeviceToDevice, st));
  IX_TRY(embed_mel_rows(h, h->rx + (size_t)n_prefix * D, codes, n, st));  // [start, codes[0..n-2]] at mel positions 0..n-1
  IX_TRY(forward_rows(h, slot, rows, 0, 0, st));
  IX_TRY(final_norm_rows(h, h->rx + (size_t)n_prefix * D, latent, n, st));
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

// ------------------------------------------------------------------------------------ bench hooks
extern "C" int ixtts_gpt_bench_gemv(ixtts_gpt* h, int which, int layer, int batch, void* stream) {
  NEED_READY(h, "gpt_bench_gemv");
  IX_ARG(layer >= 0 && layer < h->L && batch >= 1 && batch <= h->cfg.max_batch, "gpt_bench_gemv: bad argument");
  return do_gemv_which(h, which, layer, batch, (hipStream_t)stream);
}

extern "C" int ixtts_gpt_max_batch(void) { return MAXB; }

extern "C" double ixtts_gpt_step_bytes(const ixtts_gpt* h, int B, int S) {
  if (!h) return 0.0;
  const double D = h->D, FF = h->FF, V = h->V, L = h->L, es = (double)h->esize;
  double mat = L * (3 * D * D + D * D + 2 * D * FF) + V * D;
  double vec = L * (2 * D + 3 * D + D + 2 * D + FF + D) + 4 * D + V;
  double kv_read = (double)B * 2 * L * S * D * es;
  double kv_write = (double)B * 2 * L * D * es;
  double emb = (double)B * 2 * D * 4;
  return mat * es + vec * 4 + kv_read + kv_write + emb;
}

extern "C" int ixtts_gpt_destroy(ixtts_gpt* h) {
  if (!h) return IXTTS_OK;
  for (int k = 0; k <= NBKT; ++k) {
    for (int b = 0; b <= MAXB; ++b) {
      if (h->step_exec[b][k]) hipGraphExecDestroy(h->step_exec[b][k]);
      if (h->multi_exec[b][k]) hipGraphExecDestroy(h->multi_exec[b][k]);
    }
    for (int g = 0; g <= MAXG; ++g) {
      if (h->beam_exec[g][k]) hipGraphExecDestroy(h->beam_exec[g][k]);
      if (h->beam_multi_exec[g][k]) hipGraphExecDestroy(h->beam_multi_exec[g][k]);
    }
  }
  if (h->cap_stream) hipStreamDestroy(h->cap_stream);
  if (h->arena_borrowed) h->arena = nullptr;  // the owner frees it
  void* ptrs[] = {h->arena, h->stage, h->kc, h->vc, h->h, h->q, h->ff, h->att, h->part, h->logits, h->rowbuf, h->cur_len, h->gen_count,
                  h->prompt_len, h->valid_from, h->finished, h->forced, h->tokens, h->seen, h->d_samp, h->probs, h->scratch, h->beam_scores, h->hyp_score, h->hyp_worst, h->beam_src, h->hyp_len,
                  h->n_hyp, h->beam_done, h->beam_forced_flag, h->beam_forced, h->hyp_tok, h->beam_cand_v, h->beam_cand_i, h->beam_cand_n, h->beam_lcp, h->beam_stream,
                  h->rx, h->rxn, h->rq, h->ratt, h->rff, h->h2, h->wprx, h->mlp_part, h->mlp_ctr};
  for (void* p : ptrs)
    if (p) hipFree(p);
  delete h;
  return IXTTS_OK;
}

#ifdef IXTTS_TRACE
// developer timeline hooks (trace builds only; not part of include/ixtts_hip.h)
extern "C" int ixtts_trace_begin(void) {  // allocate (first call) and clear the record table
  unsigned long long* buf = nullptr;
  const size_t bytes = (size_t)TRACE_SLOTS * TRACE_WGS * 64;
  IX_HIP(hipMemcpyFromSymbol(&buf, HIP_SYMBOL(g_trace), sizeof(buf)));
  if (!buf) {
    IX_HIP(hipMalloc(&buf, bytes));
    IX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)));
  }
  IX_HIP(hipMemset(buf, 0, bytes));
  IX_HIP(hipDeviceSynchronize());
  return IXTTS_OK;
}
extern "C" int ixtts_trace_read(unsigned long long* host) {  // host: TRACE_SLOTS * TRACE_WGS * 8 u64
  IX_HIP(hipDeviceSynchronize());
  unsigned long long* buf = nullptr;
  IX_HIP(hipMemcpyFromSymbol(&buf, HIP_SYMBOL(g_trace), sizeof(buf)));
  IX_HIP(hipMemcpy(host, buf, (size_t)TRACE_SLOTS * TRACE_WGS * 64, hipMemcpyDeviceToHost));
  return IXTTS_OK;
}
#endif
