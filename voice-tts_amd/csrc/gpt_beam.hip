// gpt_beam.hip -- on-device beam-sample (the served default: num_beams=3, do_sample=True).
//
// Restates, per decode step and without touching the host:
//   indextts/gpt/transformers_generation_utils.py:3473-3543  log_softmax -> processors (min_tokens_to_keep = 2)
//        -> + beam score -> joint softmax over num_beams*V -> multinomial(2*num_beams) w/o replacement -> sort
//   indextts/gpt/transformers_beam_search.py:215-318          BeamSearchScorer.process
//   indextts/gpt/transformers_beam_search.py:930-1013         BeamHypotheses.add / is_done (any length_penalty, early_stopping False)
//   indextts/gpt/model_v2.py:199-212                          _reorder_cache (index_select of every K/V)
// `finalize` (transformers_beam_search.py:320-417) runs on the host in ixtts_gpt_beam_read.
//
// Beams live in sequence slots 0..NB-1 of the engine; reordering (token histories, `seen`
// bitmaps, K/V rows of the generated positions) is done in place, element-wise: every thread
// reads its element from all source beams before writing any destination.
#include "gpt_engine.h"
#include "gpt_kernels.h"

namespace ixtts {

constexpr int BEAM_MAX = 4;
constexpr int JOINT_MAX = BEAM_MAX * SAMP_MAXK;

struct BeamArgs {
  SamplerState s;
  float* beam_scores;   // [NB]
  int* src;             // [NB] source beam of each new beam (this step)
  float* hyp_score;     // [NB]
  int* hyp_len;         // [NB]
  int32_t* hyp_tok;     // [NB][max_new]
  int* n_hyp;
  float* worst;
  int* done;
  int32_t* forced;      // [2*NB] flat picks (beam*V + token) for the next step, valid when *forced_flag != 0
  int* forced_flag;
  float* cand_v;        // [NB][SAMP_MAXK] processed scores of each beam's surviving tokens, descending
  int* cand_i;          // [NB][SAMP_MAXK] their token ids
  int* cand_n;          // [NB] how many survive TopK + TopP
  int NB;
};

// Phase A, one workgroup per beam: log_softmax -> penalty -> temperature -> TopK (min keep 2) -> TopP (min keep 2); the
// survivors go to global memory sorted by score.  (As one workgroup looping over the beams, with the TopP sums on one
// thread, the whole step took 164 us at 3 beams.)
__global__ __launch_bounds__(1024) void beam_cand_kernel(BeamArgs a) {
  __shared__ float red[16];
  __shared__ unsigned int hist[256], wtot[4];
  __shared__ unsigned int sel_prefix, sel_remaining;
  __shared__ float cand_v[SAMP_MAXK], sort_v[SAMP_MAXK], ev[SAMP_MAXK];
  __shared__ int cand_i[SAMP_MAXK], sort_i[SAMP_MAXK];
  __shared__ int cand_n;
  __shared__ float Z_s;

  const SamplerState& s = a.s;
  const int b = blockIdx.x;
  const int V = s.V;
  // every load whose address is known at entry, up front and unconditionally
  const float* lg = s.logits + (size_t)b * V;
  const uint8_t* seen = s.seen + (size_t)b * V;
  float vals[SAMP_PT];
  uint8_t sn[SAMP_PT];
#pragma unroll
  for (int i = 0; i < SAMP_PT; ++i) {
    const int v = min((int)threadIdx.x + i * 1024, V - 1);
    vals[i] = lg[v];
    sn[i] = seen[v];
  }
  const ixtts_sampler_cfg cfg = *s.cfg;
  const int done = *a.done;
  __builtin_amdgcn_sched_barrier(0);
  if (done) return;  // hypotheses complete: HF leaves the loop here; later graph replays are no-ops
  const float theta = cfg.repetition_penalty;
  const float inv_t = cfg.temperature > 0.f ? 1.0f / cfg.temperature : 1.0f;
  {
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SAMP_PT; ++i) {
      if ((int)threadIdx.x + i * 1024 >= V) vals[i] = -INFINITY;
      mx = fmaxf(mx, vals[i]);
    }
    mx = block_max_1024(mx, red);
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < SAMP_PT; ++i) se += (threadIdx.x + i * 1024 < V) ? expf(vals[i] - mx) : 0.f;
    const float lse = mx + logf(block_sum_1024(se, red));
#pragma unroll
    for (int i = 0; i < SAMP_PT; ++i) {
      const int v = threadIdx.x + i * 1024;
      float x = -INFINITY;
      if (v < V) {
        x = vals[i] - lse;  // log_softmax
        if (cfg.suppress_stop && v == s.stop) x = -INFINITY;
        if (sn[i] && theta != 1.0f) x = (x < 0.f) ? x * theta : x / theta;
      }
      vals[i] = x;
    }
    if (cfg.typical_mass > 0.f) {  // custom processor: after the penalty, before the warpers; min_tokens_to_keep = 2 with beams
      __shared__ TypicalScratch typ;
      typical_filter_1024<SAMP_PT>(vals, V, cfg.typical_mass, 2, typ);
    }
#pragma unroll
    for (int i = 0; i < SAMP_PT; ++i) vals[i] *= inv_t;
    const int k = min(max(cfg.top_k, 2), SAMP_MAXK);
    if (threadIdx.x == 0) {
      sel_prefix = 0u;
      sel_remaining = (unsigned int)k;
      cand_n = 0;
    }
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
      __syncthreads();
      const unsigned int prefix = sel_prefix;
      const unsigned int pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
#pragma unroll
      for (int i = 0; i < SAMP_PT; ++i) {
        const int v = threadIdx.x + i * 1024;
        const unsigned int key = f2key(vals[i]);
        hist_add_aggregated(hist, (key >> shift) & 0xffu, v < V && (key & pmask) == prefix);
      }
      __syncthreads();
      radix_pick_bin(hist, wtot, &sel_prefix, &sel_remaining, shift);
    }
    const unsigned int thr = sel_prefix;
#pragma unroll
    for (int i = 0; i < SAMP_PT; ++i) {
      const int v = threadIdx.x + i * 1024;
      if (v < V && f2key(vals[i]) >= thr && vals[i] > -INFINITY) {
        const int pos = atomicAdd(&cand_n, 1);
        if (pos < SAMP_MAXK) {
          cand_v[pos] = vals[i];
          cand_i[pos] = v;
        }
      }
    }
    __syncthreads();
    const int n = min(cand_n, SAMP_MAXK);
    if (threadIdx.x < n) {
      const float mv = cand_v[threadIdx.x];
      const int mi = cand_i[threadIdx.x];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (cand_v[j] > mv || (cand_v[j] == mv && cand_i[j] < mi)) ? 1 : 0;
      sort_v[rank] = mv;
      sort_i[rank] = mi;
    }
    __syncthreads();
    // TopP: the exponentials and quotients in parallel, the two running sums in the reference's order on one thread
    if (threadIdx.x < n) ev[threadIdx.x] = expf(sort_v[threadIdx.x] - sort_v[0]);
    __syncthreads();
    if (threadIdx.x == 0) {
      float Z = 0.f;
      for (int r = 0; r < n; ++r) Z += ev[r];
      Z_s = Z;
    }
    __syncthreads();
    if (threadIdx.x < n) ev[threadIdx.x] = ev[threadIdx.x] / Z_s;
    __syncthreads();
    if (threadIdx.x == 0) {
      int keep = n;
      if (cfg.top_p < 1.0f && n > 0) {
        float tail = 0.f;
        for (int r = n - 1; r >= 2; --r) {  // never remove the top min_tokens_to_keep = 2
          tail += ev[r];
          if (tail <= 1.0f - cfg.top_p) keep = r;
          else break;
        }
      }
      a.cand_n[b] = keep;
    }
    if (threadIdx.x < n) {
      a.cand_v[b * SAMP_MAXK + threadIdx.x] = sort_v[threadIdx.x];
      a.cand_i[b * SAMP_MAXK + threadIdx.x] = sort_i[threadIdx.x];
    }
  }
}

// Phase B, one workgroup: joint draw over the beams' survivors, BeamSearchScorer.process, histories, embeddings.
__global__ __launch_bounds__(1024) void beam_step_kernel(BeamArgs a) {
  __shared__ float j_score[JOINT_MAX], j_key[JOINT_MAX];
  __shared__ int j_flat[JOINT_MAX];
  __shared__ int j_n;
  __shared__ float pick_score[2 * BEAM_MAX];
  __shared__ int pick_flat[2 * BEAM_MAX], pick_sorted[2 * BEAM_MAX];
  __shared__ float nb_score[BEAM_MAX];
  __shared__ int nb_tok[BEAM_MAX], nb_src[BEAM_MAX];
  __shared__ int act;  // 0: frozen (already done), 1: step taken, 2: became done this step

  const SamplerState& s = a.s;
  const int NB = a.NB;
  const ixtts_sampler_cfg cfg = *s.cfg;
  const int V = s.V;
  if (threadIdx.x == 0) act = (*a.done) ? 0 : 1;
  __syncthreads();
  if (act == 0) {  // hypotheses complete: HF leaves the loop here; later graph replays are no-ops
    if (threadIdx.x < NB) {
      a.src[threadIdx.x] = threadIdx.x;
      s.finished[threadIdx.x] = 1;
    }
    return;
  }
  const int kstep = s.gen_count[0] + 1;  // this step appends the k-th generated token
  {
    int off = 0;
    for (int b = 0; b < NB; ++b) {
      const int keep = min(a.cand_n[b], SAMP_MAXK);
      const float bs = a.beam_scores[b];
      if ((int)threadIdx.x < keep) {
        j_score[off + threadIdx.x] = a.cand_v[b * SAMP_MAXK + threadIdx.x] + bs;  // fp32 add, as next_token_scores_processed + beam_scores
        j_flat[off + threadIdx.x] = b * V + a.cand_i[b * SAMP_MAXK + threadIdx.x];
      }
      off += keep;
    }
    if (threadIdx.x == 0) j_n = off;
  }
  __syncthreads();

  // ---- joint multinomial(2*NB) without replacement (Gumbel top-k == p / Exp(1) top-k), or the forced draws
  const int n_tot = j_n;
  const int n_pick = 2 * NB;
  const bool forced = *a.forced_flag != 0;
  if (!forced) {
    if (threadIdx.x < n_tot) {
      float u = uniform01(cfg.seed, (unsigned int)threadIdx.x, (unsigned int)kstep);
      u = fminf(fmaxf(u, 1e-7f), 1.0f - 1e-7f);
      j_key[threadIdx.x] = j_score[threadIdx.x] - logf(-logf(u));
    }
    __syncthreads();
    if (threadIdx.x < n_tot) {
      const float mk = j_key[threadIdx.x];
      int rank = 0;
      for (int j = 0; j < n_tot; ++j) rank += (j_key[j] > mk || (j_key[j] == mk && j < (int)threadIdx.x)) ? 1 : 0;
      if (rank < n_pick) {
        pick_score[rank] = j_score[threadIdx.x];
        pick_flat[rank] = j_flat[threadIdx.x];
      }
    }
  } else if (threadIdx.x < n_pick) {
    const int f = a.forced[threadIdx.x];
    float sc = -INFINITY;
    for (int j = 0; j < n_tot; ++j)
      if (j_flat[j] == f) sc = j_score[j];
    pick_score[threadIdx.x] = sc;
    pick_flat[threadIdx.x] = f;
  }
  __syncthreads();
  // sort the draws by score, descending (stable in draw order)
  if (threadIdx.x < n_pick) {
    const float ms = pick_score[threadIdx.x];
    int rank = 0;
    for (int j = 0; j < n_pick; ++j) rank += (pick_score[j] > ms || (pick_score[j] == ms && j < (int)threadIdx.x)) ? 1 : 0;
    pick_sorted[rank] = threadIdx.x;
  }
  __syncthreads();

  // ---- BeamSearchScorer.process (one thread)
  if (threadIdx.x == 0) {
    *a.forced_flag = 0;
    int filled = 0;
    const int gen_len = kstep;  // cur_len - decoder_prompt_len
    const float lp = cfg.length_penalty;
    float best = -INFINITY;
    for (int r = 0; r < n_pick; ++r) best = fmaxf(best, pick_score[pick_sorted[r]]);
    for (int r = 0; r < n_pick && filled < NB; ++r) {
      const int pi = pick_sorted[r];
      const int flat = pick_flat[pi];
      const int bi = flat / V, tok = flat - bi * V;
      const float raw_sc = pick_score[pi];
      float sc = raw_sc;
      if (tok == s.stop) {
        if (r >= NB) continue;
        // BeamHypotheses.add(input_ids[beam].clone(), sum_logprobs, generated_len): score = sum_logprobs / generated_len ** length_penalty
        if (lp != 0.f) sc = raw_sc / powf((float)gen_len, lp);
        int nh = *a.n_hyp;
        if (nh < NB || sc > *a.worst) {
          int dst = nh;
          if (nh >= NB) {  // evict the worst, then the new worst is the second worst of the (NB+1) set
            int wi = 0;
            for (int i = 1; i < NB; ++i)
              if (a.hyp_score[i] < a.hyp_score[wi]) wi = i;
            // sorted([(s, idx)]) removes the lowest score (which may be the new one only if it was admitted: sc > worst)
            dst = wi;
          }
          a.hyp_score[dst] = sc;
          a.hyp_len[dst] = gen_len - 1;
          const int32_t* srct = s.tokens + (size_t)bi * s.max_new;
          int32_t* dstt = a.hyp_tok + (size_t)dst * s.max_new;
          for (int i = 0; i < gen_len - 1; ++i) dstt[i] = srct[i];
          if (nh < NB) {
            *a.n_hyp = nh + 1;
            *a.worst = fminf(sc, *a.worst);
          } else {
            float w = a.hyp_score[0];
            for (int i = 1; i < NB; ++i) w = fminf(w, a.hyp_score[i]);
            *a.worst = w;
          }
        }
      } else {
        nb_score[filled] = sc;
        nb_tok[filled] = tok;
        nb_src[filled] = bi;
        ++filled;
      }
    }
    // (fewer than NB non-eos candidates cannot happen: every beam keeps >= 2 tokens, at most one of them eos)
    for (; filled < NB; ++filled) {
      nb_score[filled] = -1e9f;
      nb_tok[filled] = s.stop;
      nb_src[filled] = 0;
    }
    // is_done(best_sum_logprobs = max of the draws): enough hypotheses and none attainable is better than the worst
    bool d = false;
    if (*a.n_hyp >= NB) d = (*a.worst >= (lp != 0.f ? best / powf((float)gen_len, lp) : best));
    if (d) {
      *a.done = 1;
      act = 2;
    }
  }
  __syncthreads();

  // ---- input_ids = cat(input_ids[beam_idx], tokens); `seen` follows; K/V reorder runs in the next kernel
  {
    const int k = kstep;
    for (int i = threadIdx.x; i < k - 1; i += 1024) {
      int32_t v[BEAM_MAX];
      for (int b = 0; b < NB; ++b) v[b] = s.tokens[(size_t)b * s.max_new + i];
      for (int j = 0; j < NB; ++j) s.tokens[(size_t)j * s.max_new + i] = v[nb_src[j]];
    }
    for (int v = threadIdx.x; v < V; v += 1024) {
      uint8_t sv[BEAM_MAX];
      for (int b = 0; b < NB; ++b) sv[b] = s.seen[(size_t)b * V + v];
      for (int j = 0; j < NB; ++j) s.seen[(size_t)j * V + v] = (uint8_t)(sv[nb_src[j]] | (v == nb_tok[j] ? 1 : 0));
    }
    if (threadIdx.x < NB) {
      const int j = threadIdx.x;
      if (k <= s.max_new) s.tokens[(size_t)j * s.max_new + k - 1] = nb_tok[j];
      a.beam_scores[j] = nb_score[j];
      a.src[j] = nb_src[j];
      s.gen_count[j] = k;
      s.cur_len[j] = s.prompt_len[j] + k - 1;
      if (act == 2) s.finished[j] = 1;
    }
    // embed each new beam's token: mel_embedding[tok] + mel_pos_embedding[k + 1]
    const int pos = min(k + 1, s.n_pos - 1);
    for (int j = 0; j < NB; ++j) {
      const float* e = s.mel_emb + (size_t)nb_tok[j] * s.D;
      const float* pe = s.mel_pos + (size_t)pos * s.D;
      float* h = s.h + (size_t)j * s.D;
      for (int i = threadIdx.x; i < s.D; i += 1024) h[i] = e[i] + pe[i];
    }
  }
}

constexpr int REORDER_CPT = 4;

// _reorder_cache: rows [prompt_len, cur_len) of every layer's K and V follow `src` (the prompt rows are
// identical across beams).  grid (chunks, H, L*2); one 16-byte chunk per thread, all beams read before any write.
__global__ __launch_bounds__(256) void beam_reorder_kv_kernel(void* kc, void* vc, const int* src, const int* prompt_len,
                                                               const int* cur_len, const int* done, int NB, int H, int smax,
                                                               size_t layer_stride_bytes, size_t slot_stride_bytes, int row_bytes) {
  if (*done) return;
  bool ident = true;
  for (int j = 0; j < NB; ++j) ident = ident && (src[j] == j);
  if (ident) return;
  const int p0 = prompt_len[0];
  const int rows = cur_len[0] - p0;  // generated rows already in the cache
  const int cpr = row_bytes / 16;
  const int layer = blockIdx.z >> 1, is_v = blockIdx.z & 1, hh = blockIdx.y;
  char* base0 = (char*)(is_v ? vc : kc) + layer * layer_stride_bytes + ((size_t)hh * smax + p0) * row_bytes;
  int sj[BEAM_MAX];
  for (int j = 0; j < NB; ++j) sj[j] = src[j];
  // REORDER_CPT chunks per thread (fewer, fatter workgroups: most of a 2048-row grid used to exit at once)
#pragma unroll
  for (int c = 0; c < REORDER_CPT; ++c) {
    const int idx = (blockIdx.x * REORDER_CPT + c) * 256 + threadIdx.x;
    if (idx >= rows * cpr) break;
    char* base = base0 + (size_t)idx * 16;
    uint4 v[BEAM_MAX];
    for (int b = 0; b < NB; ++b) v[b] = *reinterpret_cast<const uint4*>(base + b * slot_stride_bytes);
    for (int j = 0; j < NB; ++j)
      if (sj[j] != j) *reinterpret_cast<uint4*>(base + j * slot_stride_bytes) = v[sj[j]];
  }
}

void launch_beam_step(ixtts_gpt* h, const SamplerState& s, hipStream_t st) {
  BeamArgs a;
  a.s = s;
  a.beam_scores = h->beam_scores;
  a.src = h->beam_src;
  a.hyp_score = h->hyp_score;
  a.hyp_len = h->hyp_len;
  a.hyp_tok = h->hyp_tok;
  a.n_hyp = h->n_hyp;
  a.worst = h->hyp_worst;
  a.done = h->beam_done;
  a.forced = h->beam_forced;
  a.forced_flag = h->beam_forced_flag;
  a.cand_v = h->beam_cand_v;
  a.cand_i = h->beam_cand_i;
  a.cand_n = h->beam_cand_n;
  a.NB = h->num_beams;
  hipLaunchKernelGGL(beam_cand_kernel, dim3(h->num_beams), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(beam_step_kernel, dim3(1), dim3(1024), 0, st, a);
  const int row_bytes = HD * (int)h->esize;
  const size_t slot_stride = (size_t)h->D * h->smax * h->esize;
  const size_t layer_stride = (size_t)h->slots * slot_stride;
  // rows to move are bounded by the context bucket the graph is captured for (the host counts the steps it issues)
  const int max_rows = h->attn_bucket < NBKT ? std::min(h->smax, attn_cover(h->attn_bucket)) : h->smax;
  dim3 grid(ceil_div(max_rows * (row_bytes / 16), 256 * REORDER_CPT), h->H, h->L * 2);
  hipLaunchKernelGGL(beam_reorder_kv_kernel, grid, dim3(256), 0, st, h->kc, h->vc, (const int*)h->beam_src, (const int*)h->prompt_len,
                     (const int*)h->cur_len, (const int*)h->beam_done, h->num_beams, h->H, h->smax, layer_stride, slot_stride, row_bytes);
}

}  // namespace ixtts
