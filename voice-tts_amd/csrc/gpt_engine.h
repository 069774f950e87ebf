// gpt_engine.h -- engine state shared by gpt_engine.hip (decode, C ABI) and gpt_rows.hip (batched rows).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace ixtts {

constexpr int MAXB = 16;       // decode slots stepped together: 1..4 on the register GEMVs, 5..16 ("wide" engines) on the matrix cores (gpt_wide.h)
constexpr int MAXB_REG = 4;
constexpr int STEPS_PER_GRAPH = 8;
// beam-sample (gpt_beam.hip): beams per group, groups stepping together (group g owns slots g*NB .. g*NB+NB-1), and the strides
// of the per-group tables
constexpr int BEAM_MAX = 4;
constexpr int MAXG = MAXB / 2;
constexpr int BEAM_LCP_STRIDE = BEAM_MAX * BEAM_MAX + BEAM_MAX;  // [BEAM_MAX][BEAM_MAX] shared leading rows, then [BEAM_MAX] first row to copy
constexpr int BEAM_FORCED_STRIDE = 2 * BEAM_MAX;

enum TKind { T_VEC = 0, T_MAT_T = 1, T_MAT_N = 2, T_EMB = 3 };

struct TDesc {
  size_t off = 0;  // byte offset in the arena
  int kind = T_VEC;
  int64_t d0 = 0, d1 = 0;  // expected shape ([d0] or [d0][d1] as in the state dict)
  size_t stage_off = 0;    // float offset in the fp32 staging arena (matrices only)
  bool set = false;
};

struct LayerOff {
  size_t ln1_w, ln1_b, wqkv, bqkv, wo, bo, ln2_w, ln2_b, wfc, bfc, wpr, bpr;
};

// split-S decode attention: NSP workgroups per (head, slot); bucket k is instantiated for IT0 = ATTN_IT[k] key blocks per
// workgroup and covers contexts up to ATTN_IT[k] * NSP * 32 keys (bf16 cache; the fp32 cache holds half as many keys per
// wave-load and doubles IT0 instead)
constexpr int ATTN_NSP = 4, NBKT = 8;
constexpr int ATTN_IT[NBKT] = {2, 4, 6, 8, 10, 12, 14, 16};
constexpr int attn_cover(int bkt) { return ATTN_IT[bkt] * ATTN_NSP * 32; }

template <int D>
struct Dims;
template <>
struct Dims<1280> {
  static constexpr int R1 = 2, R4 = 1, U_QKV = 2, U_OUT = 1, U_FC = 2, U_PR = 1, U_HEAD = 4;
  // waves per workgroup: 5 makes the 5120-row FC (4 rows/wave) and the 1280-row MLP-out (1 row/wave) exactly 256 equal
  // workgroups = one per CU (with 4, 320 workgroups put two on 64 of the CUs and those set the kernel time)
  // (twice the waves with half the rows each -- QKV 8 x 1 unit, FC 10 x 1 unit -- measured: 601.5 against 588.6 us per step
  // at B=2, 670.9 against 666.6 at B=3)
  static constexpr int W_QKV = 4, W_FC = 5, W_PR = 5;
  static constexpr bool XLDS = true;  // QKV / FC activations through LDS (see gemv_reg_kernel)
};
template <>
struct Dims<128> {
  static constexpr int R1 = 4, R4 = 1, U_QKV = 1, U_OUT = 1, U_FC = 1, U_PR = 1, U_HEAD = 1;
  static constexpr int W_QKV = 4, W_FC = 4, W_PR = 4;
  static constexpr bool XLDS = true;
};

}  // namespace ixtts

struct ixtts_gpt {
  ixtts_gpt_cfg cfg;
  int D, L, H, V, FF, slots, smax;
  size_t esize;
  uint8_t* arena = nullptr;
  size_t arena_bytes = 0;
  bool arena_borrowed = false;  // ixtts_gpt_share_arena: the arena belongs to another handle
  std::vector<ixtts::LayerOff> lo;
  size_t lnf_w, lnf_b, fn_w, fn_b, whead, bhead, mel_emb, mel_pos;
  std::map<std::string, ixtts::TDesc> tens;
  bool finalized = false;
  // state
  float *h = nullptr, *q = nullptr, *ff = nullptr, *att = nullptr, *part = nullptr, *logits = nullptr, *rowbuf = nullptr;
  float* stage = nullptr;  // fp32 [N][K] staging of every matrix until finalize folds/converts it
  size_t stage_floats = 0;
  void *kc = nullptr, *vc = nullptr;
  int *cur_len = nullptr, *gen_count = nullptr, *prompt_len = nullptr, *valid_from = nullptr, *finished = nullptr,
      *forced = nullptr;
  int32_t* tokens = nullptr;
  uint8_t* seen = nullptr;
  ixtts_sampler_cfg* d_samp = nullptr;
  float* probs = nullptr;  // [slots][V] processed probabilities of the last sampling step (parity tests)
  ixtts_sampler_cfg samp_host;
  float* scratch = nullptr;
  size_t scratch_floats = 0;
  hipStream_t cap_stream = nullptr;
  // decode graphs per (batch, attention bucket): [.][NBKT] is the any-length legacy attention kernel
  hipGraphExec_t step_exec[ixtts::MAXB + 1][ixtts::NBKT + 1] = {};   // 1 decode step
  hipGraphExec_t multi_exec[ixtts::MAXB + 1][ixtts::NBKT + 1] = {};  // STEPS_PER_GRAPH steps
  // fused MLP (mlp_fused_kernel): bf16 weights, model_dim 1280, workgroup i on XCD i % 8 (probed), IXTTS_MLP != "split"
  bool mlp_fused = false;
  void* wprx = nullptr;        // [L-1][8][1280][640] bf16: c_proj repacked per XCD
  float* mlp_part = nullptr;   // [8][slots][1280]
  unsigned* mlp_ctr = nullptr; // [L][8*32 + 32]: per-layer, per-XCD arrival counters (+ a timeout mark)
  float* h2 = nullptr;         // second residual buffer (IN_LN_PART publishes the completed stream into the other one)
  float* hc = nullptr;         // the residual buffer the launch being issued works on
  bool wide = false;       // max_batch > MAXB_REG: every decode launch of this engine takes the MFMA GEMVs, whatever n_active is
  int pf_per = 2;          // IXTTS_PF builds (A/B lever): KiB of the next launch's weights each wave fetches ahead (IXTTS_PF_KIB)
  bool attn_split = true;  // IXTTS_ATTN=legacy turns the split-S kernel off (A/B timing, fallback test)
  int attn_bucket = ixtts::NBKT;  // bucket the graph being captured is built for
  int host_prompt_len[ixtts::MAXB + 2];
  int host_gen_est[ixtts::MAXB + 2];
  // beam-sample state (gpt_beam.hip): group g's beams occupy slots g*num_beams .. g*num_beams+num_beams-1; per-slot arrays
  // are indexed by the slot, per-group scalars by the group
  int num_beams = 0;
  int beam_groups = 1;                            // groups the launch being issued / captured steps
  bool group_live[ixtts::MAXG] = {};              // begun and not parked: its context counts for buckets and overflow checks
  bool group_begun[ixtts::MAXG] = {};             // device state initialised at least once
  unsigned long long* beam_stream = nullptr;      // [MAXG] RNG stream per group
  float *beam_scores = nullptr, *hyp_score = nullptr, *hyp_worst = nullptr;
  int *beam_src = nullptr, *hyp_len = nullptr, *n_hyp = nullptr, *beam_done = nullptr, *beam_forced_flag = nullptr;
  int32_t *hyp_tok = nullptr, *beam_forced = nullptr;
  float* beam_cand_v = nullptr;  // [MAXB][SAMP_MAXK] per-beam survivors of a step (gpt_beam.hip)
  int *beam_cand_i = nullptr, *beam_cand_n = nullptr;
  bool beam_every_row = false;  // IXTTS_BEAM_REORDER=full: move every generated row at each reorder (A/B test of the shared-prefix bookkeeping)
  int* beam_lcp = nullptr;  // [MAXB*MAXB] leading generated K/V rows known identical between two beams' slots, then [MAXB] first row to copy
  hipGraphExec_t beam_exec[ixtts::MAXG + 1][ixtts::NBKT + 1] = {}, beam_multi_exec[ixtts::MAXG + 1][ixtts::NBKT + 1] = {};  // per (groups, bucket)
  int beam_exec_nb = 0;
  // batched-rows workspace (prefill / latent): [max_seq][D] x4 + [max_seq][4D]
  float *rx = nullptr, *rxn = nullptr, *rq = nullptr, *ratt = nullptr, *rff = nullptr;
};

#define A_F32(off) reinterpret_cast<float*>(h->arena + (off))
#define A_PTR(off) reinterpret_cast<void*>(h->arena + (off))



namespace ixtts {
// Batched causal pass of T rows through the 24 layers for sequence slot `slot`:
// rows X [T][D] (h->rx, in place) sit at cache positions pos0..pos0+T-1; keys < valid_from are masked.
int forward_rows(ixtts_gpt* h, int slot, int T, int pos0, int valid_from, hipStream_t st);
}  // namespace ixtts

namespace ixtts {
int final_norm_rows(ixtts_gpt* h, const float* x, float* y, int T, hipStream_t st);
int embed_mel_rows(ixtts_gpt* h, float* x, const int32_t* codes, int rows, hipStream_t st);
struct SamplerState;
void launch_beam_step(ixtts_gpt* h, const SamplerState& s, hipStream_t st);
}  // namespace ixtts
