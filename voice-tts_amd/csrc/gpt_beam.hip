// gpt_beam.hip -- on-device beam-sample (the served default: num_beams=3, do_sample=True) and, with
// do_sample == 0, beam search proper (the joint top 2 * num_beams instead of a multinomial draw).
//
// Restates, per decode step and without touching the host:
//   indextts/gpt/transformers_generation_utils.py:3473-3543  log_softmax -> processors (min_tokens_to_keep = 2)
//        -> + beam score -> joint softmax over num_beams*V -> multinomial(2*num_beams) w/o replacement -> sort
//   indextts/gpt/transformers_beam_search.py:215-318          BeamSearchScorer.process
//   indextts/gpt/transformers_beam_search.py:930-1013         BeamHypotheses.add / is_done (any length_penalty, early_stopping False)
//   indextts/gpt/model_v2.py:199-212                          _reorder_cache (index_select of every K/V)
// `finalize` (transformers_beam_search.py:320-417) runs on the host in ixtts_gpt_beam_read.
//
// A beam GROUP (the beams of one prompt) lives in sequence slots g*NB .. g*NB+NB-1 of the engine; reordering (token
// histories, `seen` bitmaps, K/V rows of the generated positions) is done in place, element-wise: every thread reads its
// element from all source beams before writing any destination.  Several groups -- the text segments of one request, or of
// several requests (infer_v2.py:616 decodes them one after another) -- step together: every kernel below takes the group
// from its block index and works on that group's slots and that group's scorer state only, so a group's tokens do not
// depend on its company (tests/test_gpu_beam_groups.py).
#include "gpt_engine.h"
#include "gpt_kernels.h"

namespace ixtts {

static_assert(BEAM_MAX == 4, "pick4 / the 16-entry lcp table below");

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
  int* lcp;             // [BEAM_MAX][BEAM_MAX] leading generated K/V rows two slots share, then [BEAM_MAX] first row each slot must take from its source
  const unsigned long long* stream;  // [G] RNG stream of each group (0: the plain seed)
  int NB;
  int G;                // groups stepping together; group g owns slots g*NB.., scalars at [g], per-slot arrays at [g*NB + beam]
};

// The arguments as group g sees them: per-slot arrays start at its first slot, per-group scalars at its entry.
__device__ __forceinline__ BeamArgs beam_group_view(const BeamArgs& in, int g) {
  BeamArgs a = in;
  const size_t sb = (size_t)g * in.NB;
  SamplerState& s = a.s;
  s.logits += sb * s.V;
  s.seen += sb * s.V;
  s.tokens += sb * s.max_new;
  s.gen_count += sb;
  s.cur_len += sb;
  s.prompt_len += sb;
  s.finished += sb;
  s.h += sb * s.D;
  a.beam_scores += sb;
  a.src += sb;
  a.hyp_score += sb;
  a.hyp_len += sb;
  a.hyp_tok += sb * s.max_new;
  a.n_hyp += g;
  a.worst += g;
  a.done += g;
  a.forced += (size_t)g * BEAM_FORCED_STRIDE;
  a.forced_flag += g;
  a.cand_v += sb * SAMP_MAXK;
  a.cand_i += sb * SAMP_MAXK;
  a.cand_n += sb;
  a.lcp += (size_t)g * BEAM_LCP_STRIDE;
  a.stream += g;
  return a;
}

// Phase A, one workgroup per beam: log_softmax -> penalty -> temperature -> TopK (min keep 2) -> TopP (min keep 2); the
// survivors go to global memory sorted by score.  (As one workgroup looping over the beams, with the TopP sums on one
// thread, the whole step took 164 us at 3 beams.)
__global__ __launch_bounds__(1024) void beam_cand_kernel(BeamArgs a) {
  __shared__ float red[16];
  __shared__ TopkScratch tk;
  __shared__ float sort_v[SAMP_MAXK], ev[SAMP_MAXK], qv[SAMP_MAXK];
  __shared__ int sort_i[SAMP_MAXK];

  DBG_TS(0);
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
  const int done = a.done[b / a.NB];  // (b is the engine slot: beam b % NB of group b / NB)
  __builtin_amdgcn_sched_barrier(0);
  if (done) return;  // hypotheses complete: HF leaves the loop here; later graph replays are no-ops
  DBG_TS(1);
  const float theta = cfg.repetition_penalty;
  // do_sample == 0: beam search proper (`_beam_search`'s topk branch, generation_utils.py:3520-3524) -- the processors run, the
  // warpers (temperature / top-k / top-p) are sampling-only (:1020); a beam's 2 * NB best are all the joint top 2 * NB can hold
  const bool sampling = cfg.do_sample != 0;
  const float inv_t = sampling && cfg.temperature > 0.f ? 1.0f / cfg.temperature : 1.0f;
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
    DBG_TS(2);
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
    const int k = sampling ? min(max(cfg.top_k, 2), SAMP_MAXK) : 2 * a.NB;
    DBG_TS(3);
    const int n = topk_sorted_1024<SAMP_PT>(vals, V, k, tk, sort_v, sort_i);
    DBG_TS(4);
    if (threadIdx.x < 64) {  // TopP (never removes the top min_tokens_to_keep = 2) and the survivors' way out, by wave 0
      float Z;
      const int keep = sampling ? topp_wave0(sort_v, n, cfg.top_p, 2, ev, qv, &Z) : n;
      const int lane = threadIdx.x;
      if (lane == 0) a.cand_n[b] = keep;
      if (lane < n) {
        a.cand_v[b * SAMP_MAXK + lane] = sort_v[lane];
        a.cand_i[b * SAMP_MAXK + lane] = sort_i[lane];
      }
      if (lane + 64 < n) {
        a.cand_v[b * SAMP_MAXK + lane + 64] = sort_v[lane + 64];
        a.cand_i[b * SAMP_MAXK + lane + 64] = sort_i[lane + 64];
      }
      DBG_TS(5);
    }
  }
}

// Phase B, one workgroup: joint draw over the beams' survivors, BeamSearchScorer.process, histories, embeddings.
// What shapes it (in-kernel timestamps, tools/beam_dbg.py): one wave runs ~0.6 instructions per ns and a dependent LDS read
// costs ~100 cycles, so per-survivor arithmetic (the Gumbel keys) goes one survivor per thread, rank loops read LDS four
// keys at a time, and the strictly serial scorer is a short scalar loop on lane 0 over LDS arrays -- not wave-wide
// reductions or unrolled selects over padded register arrays (16 us that way, 20 us as eight barrier-separated phases).
// Everything whose address is known at entry is loaded up front; waves 1..15 fetch the rows they will permute (token
// histories, `seen` flags) while wave 0 runs the scorer.
__global__ __launch_bounds__(1024) void beam_step_kernel(BeamArgs a_all) {
  const BeamArgs a = beam_group_view(a_all, blockIdx.x);  // one workgroup per group
  __shared__ alignas(16) float j_key[JOINT_MAX + 4];
  __shared__ float pick_score[2 * BEAM_MAX], q_sc[2 * BEAM_MAX], nb_score[BEAM_MAX], hs_s[BEAM_MAX];
  __shared__ int pick_tok[2 * BEAM_MAX], pick_beam[2 * BEAM_MAX], q_tok[2 * BEAM_MAX], q_beam[2 * BEAM_MAX], forced_s[2 * BEAM_MAX];
  __shared__ int nb_tok[BEAM_MAX], nb_src[BEAM_MAX], add_dst[BEAM_MAX], add_src[BEAM_MAX];
  __shared__ int lcp_s[BEAM_MAX * BEAM_MAX];
  __shared__ int act, n_add;  // act 1: step taken, 2: became done this step

  DBG_TS(10);
  const SamplerState& s = a.s;
  const int NB = a.NB;
  const int V = s.V;
  const int t = threadIdx.x;
  const ixtts_sampler_cfg cfg = *s.cfg;
  const int st_done = *a.done, st_gen = s.gen_count[0], st_nhyp = *a.n_hyp;
  const bool forced = *a.forced_flag != 0;
  const float st_worst = *a.worst;
  const int st_prompt = s.prompt_len[min(t, NB - 1)];
  const int st_lcp = a.lcp[t & (BEAM_MAX * BEAM_MAX - 1)];
  const int st_pick = a.forced[min(t, 2 * NB - 1)];
  const float st_hs = a.hyp_score[min(t, NB - 1)];
  const unsigned long long st_stream = *a.stream;
  // survivor (beam t / 128, rank t % 128) of this thread
  const int my_b = min(t >> 7, NB - 1), my_r = t & (SAMP_MAXK - 1);
  const float my_v = a.cand_v[my_b * SAMP_MAXK + my_r], my_bs = a.beam_scores[my_b];
  const int my_i = a.cand_i[my_b * SAMP_MAXK + my_r];
  int c_n[BEAM_MAX];
#pragma unroll
  for (int b = 0; b < BEAM_MAX; ++b) c_n[b] = a.cand_n[min(b, NB - 1)];
  __builtin_amdgcn_sched_barrier(0);
  if (st_done) {  // hypotheses complete: HF leaves the loop here; later graph replays are no-ops
    if (t < NB) {
      a.src[t] = t;
      s.finished[t] = 1;
    }
    return;
  }
  DBG_TS(11);
  const int kstep = st_gen + 1;  // this step appends the k-th generated token
  // embed row of the next position: mel_pos_embedding[k + 1]
  const int pos = min(kstep + 1, s.n_pos - 1);
  const float* pe = s.mel_pos + (size_t)pos * s.D;
  float pe0 = 0.f, pe1 = 0.f;
  if (t < s.D) pe0 = pe[t];
  if (t + 1024 < s.D) pe1 = pe[t + 1024];
  if (t < BEAM_MAX * BEAM_MAX) lcp_s[t] = st_lcp;
  if (t < BEAM_MAX) hs_s[t] = st_hs;
  const int n_pick = 2 * NB;
  if (t < 2 * BEAM_MAX) {
    forced_s[t] = st_pick;
    pick_score[t] = -INFINITY;
    pick_tok[t] = forced ? st_pick % V : 0;
    pick_beam[t] = forced ? st_pick / V : 0;
  }
  // ---- the joint list: survivor (b, r) sits at q = (survivors of the beams before b) + r
  int off = 0, n_tot = 0;
#pragma unroll
  for (int b = 0; b < BEAM_MAX; ++b) {
    const int keep = b < NB ? min(c_n[b], SAMP_MAXK) : 0;
    if (b == (t >> 7)) off = n_tot;
    n_tot += keep;
  }
  const bool valid = t < BEAM_MAX * SAMP_MAXK && (t >> 7) < NB && my_r < min(c_n[my_b], SAMP_MAXK);
  const int q = off + my_r;
  const float score = my_v + my_bs;  // fp32 add, as next_token_scores_processed + beam_scores
  float key = -INFINITY;
  if (valid && !forced) {
    // joint multinomial(2*NB) without replacement == the 2*NB largest of score + Gumbel noise (p / Exp(1) top-k)
    float u = uniform01(cfg.seed + st_stream * 0xD1B54A32D192ED03ull, (unsigned int)q, (unsigned int)kstep);
    u = fminf(fmaxf(u, 1e-7f), 1.0f - 1e-7f);
    key = cfg.do_sample ? score - logf(-logf(u)) : score;  // do_sample == 0: `torch.topk` of the joint scores, ties to the lower flat index
    j_key[q] = key;
  }
  if (t < 4) j_key[n_tot + t] = -INFINITY;  // the rank loop below reads four keys at a time
  __syncthreads();
  DBG_TS(12);
  if (valid) {
    if (!forced) {
      int rank = 0;
      for (int j = 0; j < n_tot; j += 4) {
        const float4 k4 = *reinterpret_cast<const float4*>(j_key + j);
        rank += (k4.x > key || (k4.x == key && j < q)) ? 1 : 0;
        rank += (k4.y > key || (k4.y == key && j + 1 < q)) ? 1 : 0;
        rank += (k4.z > key || (k4.z == key && j + 2 < q)) ? 1 : 0;
        rank += (k4.w > key || (k4.w == key && j + 3 < q)) ? 1 : 0;
      }
      if (rank < n_pick) {
        pick_score[rank] = score;
        pick_tok[rank] = my_i;
        pick_beam[rank] = my_b;
      }
    } else {
      const int flat = my_b * V + my_i;
      for (int d = 0; d < n_pick; ++d)
        if (forced_s[d] == flat) pick_score[d] = score;
    }
  }
  __syncthreads();
  DBG_TS(13);

  // waves 1..15 hold the rows they will permute (token histories, `seen` flags as one bit per beam), loaded under wave 0's work
  constexpr int NPERM = 1024 - 64, HCH = 3;
  const int tp = t - 64;
  int32_t hist[HCH][BEAM_MAX];
  unsigned int svp[SAMP_PT];
  if (t < 64) {
    // ---- the draws sorted by score, descending (stable in draw order): lane d places draw d
    if (t < n_pick) {
      float ps[2 * BEAM_MAX];
#pragma unroll
      for (int j = 0; j < 2 * BEAM_MAX; ++j) ps[j] = pick_score[j];
      float mine = 0.f;
#pragma unroll
      for (int j = 0; j < 2 * BEAM_MAX; ++j) mine = j == t ? ps[j] : mine;
      int rank = 0;
#pragma unroll
      for (int j = 0; j < 2 * BEAM_MAX; ++j) rank += (j < n_pick && (ps[j] > mine || (ps[j] == mine && j < t))) ? 1 : 0;
      q_sc[rank] = mine;
      q_tok[rank] = pick_tok[t];
      q_beam[rank] = pick_beam[t];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    DBG_TS(14);
    // ---- BeamSearchScorer.process (lane 0)
    if (t == 0) {
      if (forced) *a.forced_flag = 0;
      int filled = 0, nh = st_nhyp, adds = 0;
      float worst = st_worst;
      const int gen_len = kstep;  // cur_len - decoder_prompt_len
      const float lp = cfg.length_penalty;
      const float best = q_sc[0];  // max of the draws
      for (int r = 0; r < n_pick && filled < NB; ++r) {
        const int bi = q_beam[r], tok = q_tok[r];
        float sc = q_sc[r];
        if (tok == s.stop) {
          if (r >= NB) continue;
          // BeamHypotheses.add(input_ids[beam].clone(), sum_logprobs, generated_len): score = sum_logprobs / generated_len ** length_penalty
          if (lp != 0.f) sc = sc / powf((float)gen_len, lp);
          if (nh < NB || sc > worst) {
            int dst = nh;
            if (nh >= NB) {  // evict the worst: sorted([(s, idx)]) removes the lowest score
              dst = 0;
              for (int i = 1; i < NB; ++i)
                if (hs_s[i] < hs_s[dst]) dst = i;
            }
            hs_s[dst] = sc;
            a.hyp_score[dst] = sc;
            a.hyp_len[dst] = gen_len - 1;
            add_dst[adds] = dst;  // the tokens are copied below, by every thread, from the histories held in registers
            add_src[adds] = bi;
            ++adds;
            if (nh < NB) {
              ++nh;
              worst = fminf(sc, worst);
            } else {
              worst = hs_s[0];
              for (int i = 1; i < NB; ++i) worst = fminf(worst, hs_s[i]);
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
      if (adds) {
        *a.n_hyp = nh;
        *a.worst = worst;
      }
      n_add = adds;
      // is_done(best_sum_logprobs = max of the draws): enough hypotheses and none attainable is better than the worst
      const bool now_done = nh >= NB && worst >= (lp != 0.f ? best / powf((float)gen_len, lp) : best);
      if (now_done) *a.done = 1;
      act = now_done ? 2 : 1;
    }
    DBG_TS(15);
  } else {
    uint8_t sv[BEAM_MAX][SAMP_PT];
#pragma unroll
    for (int b = 0; b < BEAM_MAX; ++b) {
      const int bb = min(b, NB - 1);
#pragma unroll
      for (int c = 0; c < HCH; ++c) hist[c][b] = s.tokens[(size_t)bb * s.max_new + min(tp + c * NPERM, s.max_new - 1)];
#pragma unroll
      for (int i = 0; i < SAMP_PT; ++i) sv[b][i] = s.seen[(size_t)bb * V + min(tp + i * NPERM, V - 1)];
    }
#pragma unroll
    for (int i = 0; i < SAMP_PT; ++i) {
      svp[i] = 0u;
#pragma unroll
      for (int b = 0; b < BEAM_MAX; ++b) svp[i] |= (sv[b][i] ? 1u : 0u) << b;
    }
    DBG_TSW(20, 1);
  }
  __syncthreads();
  DBG_TS(16);

  // ---- input_ids = cat(input_ids[beam_idx], tokens); `seen` follows; K/V reorder runs in the next kernel
  const int k = kstep;
  int sj[BEAM_MAX], tj[BEAM_MAX];
#pragma unroll
  for (int j = 0; j < BEAM_MAX; ++j) {
    sj[j] = nb_src[min(j, NB - 1)];
    tj[j] = nb_tok[min(j, NB - 1)];
  }
  // the chosen tokens' embedding rows: the one dependent round trip of this kernel, issued before the stores
  float e0[BEAM_MAX], e1[BEAM_MAX];
#pragma unroll
  for (int j = 0; j < BEAM_MAX; ++j) {
    const float* e = s.mel_emb + (size_t)tj[j] * s.D;
    e0[j] = t < s.D ? e[t] : 0.f;
    e1[j] = t + 1024 < s.D ? e[t + 1024] : 0.f;
  }
  // K/V rows to move (beam_reorder_kv_kernel): slot j takes rows [lo_j, k - 1) of slot src_j -- below lo_j the two already
  // hold the same bytes (they were copied from a common ancestor); afterwards beams of one source agree on all k - 1 rows,
  // the others where their sources did
  if (t < BEAM_MAX * BEAM_MAX + BEAM_MAX) {
    const int c = kstep - 1;
    const bool is_lo = t >= BEAM_MAX * BEAM_MAX;
    const int i = is_lo ? t - BEAM_MAX * BEAM_MAX : t / BEAM_MAX, j = is_lo ? i : t % BEAM_MAX;
    if (i < NB && j < NB) {
      const int si = is_lo ? i : nb_src[i], sjj = nb_src[j];  // first row to copy: (slot i, its source); shared rows after: (source i, source j)
      a.lcp[t] = si == sjj ? c : min(lcp_s[si * BEAM_MAX + sjj], c);
    }
  }
  auto pick4 = [](const auto(&v)[BEAM_MAX], int i) { return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3]; };
  if (t >= 64) {
    const int adds = n_add;
    for (int x = 0; x < adds; ++x) {  // finished hypotheses: input_ids[beam].clone() (before the permutation below)
      int32_t* dstt = a.hyp_tok + (size_t)add_dst[x] * s.max_new;
#pragma unroll
      for (int c = 0; c < HCH; ++c)
        if (tp + c * NPERM < k - 1) dstt[tp + c * NPERM] = pick4(hist[c], add_src[x]);
    }
#pragma unroll
    for (int j = 0; j < BEAM_MAX; ++j)
      if (j < NB && sj[j] != j) {
#pragma unroll
        for (int c = 0; c < HCH; ++c)
          if (tp + c * NPERM < k - 1) s.tokens[(size_t)j * s.max_new + tp + c * NPERM] = pick4(hist[c], sj[j]);
      }
    for (int i = tp + HCH * NPERM; i < k - 1; i += NPERM) {  // (max_seq > 2880)
      int32_t v[BEAM_MAX];
#pragma unroll
      for (int b = 0; b < BEAM_MAX; ++b) v[b] = s.tokens[(size_t)min(b, NB - 1) * s.max_new + i];
      for (int x = 0; x < adds; ++x) a.hyp_tok[(size_t)add_dst[x] * s.max_new + i] = pick4(v, add_src[x]);
#pragma unroll
      for (int j = 0; j < BEAM_MAX; ++j)
        if (j < NB && sj[j] != j) s.tokens[(size_t)j * s.max_new + i] = pick4(v, sj[j]);
    }
#pragma unroll
    for (int i = 0; i < SAMP_PT; ++i) {
      const int v = tp + i * NPERM;
      if (v < V) {
#pragma unroll
        for (int j = 0; j < BEAM_MAX; ++j)
          if (j < NB) {
            const unsigned int old = (svp[i] >> sj[j]) & 1u;
            if (sj[j] != j || v == tj[j]) s.seen[(size_t)j * V + v] = (uint8_t)(old | (v == tj[j] ? 1u : 0u));
          }
      }
    }
  }
  if (t < NB) {
    const int j = t;
    if (k <= s.max_new) s.tokens[(size_t)j * s.max_new + k - 1] = nb_tok[j];
    a.beam_scores[j] = nb_score[j];
    a.src[j] = nb_src[j];
    s.gen_count[j] = k;
    s.cur_len[j] = st_prompt + k - 1;
    if (act == 2) s.finished[j] = 1;
  }
  DBG_TS(17);
  // embed each new beam's token: mel_embedding[tok] + mel_pos_embedding[k + 1]
#pragma unroll
  for (int j = 0; j < BEAM_MAX; ++j)
    if (j < NB) {
      float* h = s.h + (size_t)j * s.D;
      if (t < s.D) h[t] = e0[j] + pe0;
      if (t + 1024 < s.D) h[t + 1024] = e1[j] + pe1;
      const float* e = s.mel_emb + (size_t)tj[j] * s.D;
      for (int i = t + 2048; i < s.D; i += 1024) h[i] = e[i] + pe[i];
    }
  DBG_TS(18);
}

constexpr int REORDER_CPT = 4, REORDER_GX = 2;  // 2 x 1024 chunks = 256 rows of one head per pass

// _reorder_cache: the generated rows [prompt_len, cur_len) of every layer's K and V follow `src` (the prompt rows are
// identical across beams).  Slot j needs only rows [lo_j, cur_len) of slot src_j: below lo_j the two slots already hold the
// same bytes (beam_step_kernel tracks, per pair of slots, how many leading rows were copied from a common ancestor --
// beams re-converge every few steps, so the rows to move are the last few, not the whole history: 17 us per step on
// average, 129 us at 1000 rows, when every row was moved).  grid (REORDER_GX, H, L*2), each workgroup striding over the
// chunks from the NEWEST row backwards (the usual few rows are one pass; the grid no longer grows with the context
// bucket); one 16-byte chunk per thread and step, all beams read before any write.
__global__ __launch_bounds__(256) void beam_reorder_kv_kernel(void* kc, void* vc, const int* src, const int* lo, const int* prompt_len,
                                                               const int* cur_len, const int* done, int NB, int H, int smax,
                                                               size_t layer_stride_bytes, size_t slot_stride_bytes, int row_bytes, int every_row) {
  // grid.x = REORDER_GX workgroups per group: rebase everything on this group's first slot
  const int grp = blockIdx.x / REORDER_GX, bx = blockIdx.x % REORDER_GX;
  const size_t sb = (size_t)grp * NB;
  if (done[grp]) return;
  src += sb;
  lo += (size_t)grp * BEAM_LCP_STRIDE;
  prompt_len += sb;
  cur_len += sb;
  const int p0 = prompt_len[0];
  const int rows = cur_len[0] - p0;  // generated rows already in the cache
  int sj[BEAM_MAX], lj[BEAM_MAX];
  int lo_min = rows;
  for (int j = 0; j < BEAM_MAX; ++j) {
    sj[j] = j < NB ? src[j] : j;
    lj[j] = (j < NB && sj[j] != j) ? (every_row ? 0 : lo[j]) : rows;
    lo_min = min(lo_min, lj[j]);
  }
  const int cpr = row_bytes / 16;
  const int total = (rows - lo_min) * cpr;  // 16-byte chunks to look at, counted back from the newest row's last
  if ((int)(bx * REORDER_CPT * 256) >= total) return;
  const int layer = blockIdx.z >> 1, is_v = blockIdx.z & 1, hh = blockIdx.y;
  char* base0 = (char*)(is_v ? vc : kc) + layer * layer_stride_bytes + sb * slot_stride_bytes + ((size_t)hh * smax + p0) * row_bytes;
  for (int blk = bx; blk * REORDER_CPT * 256 < total; blk += REORDER_GX) {
#pragma unroll
    for (int c = 0; c < REORDER_CPT; ++c) {
      const int back = (blk * REORDER_CPT + c) * 256 + threadIdx.x;
      if (back >= total) break;
      const int idx = rows * cpr - 1 - back;
      const int row = idx / cpr;
      char* base = base0 + (size_t)idx * 16;
      uint4 v[BEAM_MAX];
#pragma unroll
      for (int b = 0; b < BEAM_MAX; ++b) v[b] = *reinterpret_cast<const uint4*>(base + min(b, NB - 1) * slot_stride_bytes);
#pragma unroll
      for (int j = 0; j < BEAM_MAX; ++j)
        if (j < NB && row >= lj[j]) *reinterpret_cast<uint4*>(base + j * slot_stride_bytes) = sj[j] == 0 ? v[0] : sj[j] == 1 ? v[1] : sj[j] == 2 ? v[2] : v[3];
    }
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
  a.lcp = h->beam_lcp;
  a.stream = h->beam_stream;
  a.NB = h->num_beams;
  a.G = h->beam_groups;
  hipLaunchKernelGGL(beam_cand_kernel, dim3(h->num_beams * h->beam_groups), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(beam_step_kernel, dim3(h->beam_groups), dim3(1024), 0, st, a);
  const int row_bytes = HD * (int)h->esize;
  const size_t slot_stride = (size_t)h->D * h->smax * h->esize;
  const size_t layer_stride = (size_t)h->slots * slot_stride;
  dim3 grid(REORDER_GX * h->beam_groups, h->H, h->L * 2);
  hipLaunchKernelGGL(beam_reorder_kv_kernel, grid, dim3(256), 0, st, h->kc, h->vc, (const int*)h->beam_src, (const int*)(h->beam_lcp + BEAM_MAX * BEAM_MAX), (const int*)h->prompt_len,
                     (const int*)h->cur_len, (const int*)h->beam_done, h->num_beams, h->H, h->smax, layer_stride, slot_stride, row_bytes,
                     h->beam_every_row ? 1 : 0);
}

}  // namespace ixtts

#ifdef BEAM_DBG
extern "C" int ixtts_debug_beam_ts(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ixtts::g_beam_dbg), 64 * 8);
}
#endif
