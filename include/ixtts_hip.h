/* ixtts_hip.h -- C ABI of libixtts_hip.so: the MI355X (gfx950) hot path of IndexTTS2.
 *
 * Drop-in boundary for ONE path of caishiqing/voice-tts: the autoregressive GPT decode
 * step and the BigVGAN vocoder of `indextts.infer_v2.IndexTTS2.infer()`.
 * Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * Conventions
 *   - every `*_dev` pointer is DEVICE memory on the current HIP device; `*_host` is host memory;
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = the default stream); all
 *     compute entry points are asynchronous on it unless stated otherwise;
 *   - return value: 0 on success, a negative IXTTS_ERR_* code otherwise; no C++
 *     exception crosses the ABI; `ixtts_last_error()` gives the text of the last failure
 *     on the calling thread;
 *   - handles are not re-entrant: one in-flight call per handle (the reference serialises
 *     `infer()` behind `inference_lock`, server.py:25,384).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference repo root).
 */
#ifndef IXTTS_HIP_H
#define IXTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IXTTS_OK 0
#define IXTTS_ERR_ARG (-1)   /* bad argument / shape mismatch          */
#define IXTTS_ERR_HIP (-2)   /* a HIP runtime call failed               */
#define IXTTS_ERR_STATE (-3) /* call order violated (e.g. not finalized) */
#define IXTTS_ERR_NOMEM (-4) /* device/host allocation failed           */
#define IXTTS_ERR_NAME (-5)  /* unknown tensor name                     */

const char* ixtts_version(void);
const char* ixtts_last_error(void);

/* ------------------------------------------------------------------------------------
 * Seam 3 -- fused anti-aliased SnakeBeta activation
 * replaces: `extern "C" torch::Tensor fwd_cuda(input, up_filter, down_filter, alpha, beta)`
 *   indextts/s2mel/modules/bigvgan/alias_free_activation/cuda/anti_alias_activation.cpp:19-22
 *   indextts/s2mel/modules/bigvgan/alias_free_activation/cuda/anti_alias_activation_cuda.cu:212-246
 * y[b,c,:] = Down2(SnakeBeta(Up2(x[b,c,:]))), 12-tap filters, replicate padding, log-scale
 * alpha/beta [C] (exp applied inside, .cu:86-89).  Semantics follow the reference's CPU
 * (torch) path `alias_free_activation/torch/act.py:24-30` exactly, including the sequence
 * edges.  x, y: [B,C,T] contiguous fp32; y may not alias x.  T == 0 is a no-op.
 */
int ixtts_aa_snake_f32(const float* x_dev, float* y_dev, const float* up12_dev, const float* down12_dev,
                       const float* log_alpha_dev, const float* log_beta_dev, int B, int C, int T, void* stream);

/* ------------------------------------------------------------------------------------
 * Row N1 helper -- full (unmasked, non-causal) fp32 attention of the s2mel DiT
 * replaces: `F.scaled_dot_product_attention(q, k, v, attn_mask=mask)` with an all-true mask
 *   indextts/s2mel/modules/gpt_fast/model.py:303 (called from diffusion_transformer.py:238)
 * q, k, v, out: fp32, element (b, t, h, d) at base[b*stride_b + t*stride_t + h*stride_h + d] (d contiguous,
 * head_dim 64, strides in floats and multiples of 4); out = softmax(scale * q k^T) v per (b, h).
 * workspace_dev (optional, ixtts_attn_full_workspace_bytes(B,H,T) bytes of scratch owned by the caller): holds the K / V operand
 * planes of the default kernel (csrc/attn_full_x3.hip: every fp32 product as six bf16 MFMA partial products of exactly split
 * operands, fp32 accumulate -- fp32-quality results, tests/test_gpu_s2mel.py) and the partials of the key-range split it uses when
 * the un-split grid cannot fill the GPU (+ a merge pass).  Without it, or with IXTTS_ATTN_FULL=f32 in the environment, the
 * fp32-MFMA kernel of csrc/attn_full.hip runs.
 */
size_t ixtts_attn_full_workspace_bytes(int B, int H, int T);
int ixtts_attn_full_f32(const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev, int B, int H, int T,
                        int head_dim, long stride_b, long stride_t, long stride_h, long ostride_b, long ostride_t,
                        long ostride_h, float scale, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Row N1 helpers -- the element/row chains between the DiT's GEMMs, one fp32 pass each (csrc/dit_ops.hip).
 *
 * ixtts_adaln_rmsnorm_f32  replaces AdaptiveLayerNorm.forward over RMSNorm (gpt_fast/model.py:18-37,362-372):
 *     out[b,t,:] = wb[b,H:2H] + wb[b,0:H] * (x[b,t,:] * rsqrt(mean(x^2) + eps) * g), wb = project_layer(c) [B,2H].
 * ixtts_ln_modulate_f32    replaces FinalLayer's `modulate(norm_final(x), shift, scale)` (diffusion_transformer.py:83-100):
 *     out = layer_norm(x, eps, no affine) * (1 + ss[b,H:2H]) + ss[b,0:H], ss = adaLN_modulation(c) [B,2H] (shift | scale).
 * ixtts_rope_qk_f32        replaces apply_rotary_emb on q and k (gpt_fast/model.py:289-301,348-360), in place on the
 *     wqkv output [B*T,3H]: pair i of head h at columns h*hd+2i,+1 is multiplied by cos_sin[t][i] = (cos, sin).
 * ixtts_swiglu_f32         replaces `F.silu(w1(x)) * w3(x)` (gpt_fast/model.py:316-326) on the fused [w1; w3] GEMM
 *     output u [rows,2F]: out[r,j] = silu(u[r,j]) * u[r,F+j].
 * ixtts_wn_gate_f32        replaces fused_add_tanh_sigmoid_multiply (wavenet.py:142-160): out[b,c,t] =
 *     tanh(a[b,c,t] + g[b,off+c]) * sigmoid(a[b,C+c,t] + g[b,off+C+c]); a [B,2C,T], g [B,g_stride], out [B,C,T].
 * ixtts_wn_gate_rows_f32   the same gate in row layout: a [rows,2C] -> out [rows,C], row r taking the gate biases of batch
 *     entry min(r / rows_per_batch, B-1) (the WaveNet head keeps its sequences as rows so that every conv is a plain
 *     row-major GEMM; rows between two sequences are computed and never read).
 * ixtts_reflect_halo_rows_f32  refreshes the reflect padding (SConv1d pad_mode "reflect", wavenet.py:103-140) of a row-layout
 *     buffer p [B, left+T+right, C] in place from its interior rows.
 * All tensors contiguous fp32 device memory; H, F, C multiples of 4, H <= 2048.
 */
int ixtts_adaln_rmsnorm_f32(const float* x_dev, const float* wb_dev, const float* g_dev, float* out_dev, int B, int T, int H, float eps,
                            void* stream);
int ixtts_ln_modulate_f32(const float* x_dev, const float* shift_scale_dev, float* out_dev, int B, int T, int H, float eps, void* stream);
int ixtts_rope_qk_f32(float* qkv_dev, const float* cos_sin_dev, int B, int T, int H, int head_dim, void* stream);
int ixtts_swiglu_f32(const float* u_dev, float* out_dev, long rows, int F, void* stream);
int ixtts_wn_gate_f32(const float* a_dev, const float* g_dev, float* out_dev, int B, int C, int T, long g_stride, int g_offset, void* stream);
int ixtts_wn_gate_rows_f32(const float* a_dev, const float* g_dev, float* out_dev, long rows, int C, long rows_per_batch, int B, long g_stride,
                           int g_offset, void* stream);
int ixtts_reflect_halo_rows_f32(float* p_dev, int B, int T, int C, int left, int right, void* stream);

/* Row N1 -- the linear layers of the s2mel DiT / WaveNet: C[M][N] (+)= A[M][K] . W[N][K]^T + bias, fp32 in and out
 * replaces: `nn.Linear` / 1x1 and k-tap Conv1d of indextts/s2mel/modules/gpt_fast/model.py:242-326 (wqkv, wo, w1 | w3, w2),
 *   wavenet.py:103-174 (in_layers as one row-shifted GEMM per tap, res_skip_layers), diffusion_transformer.py:186-257 (merge / skip
 *   linears), called through torch's fp32 library GEMMs in r02.
 * Arithmetic (csrc/gemm_x6.hip): every fp32 product as six exact bf16 MFMA partial products of three-way split operands, fp32
 * accumulation -- fp32-quality results (tests/test_gpu_gemm_x6.py: error against fp64 no larger than the fp32 library GEMM's).
 * Operands travel as bf16 PLANES [K/16][plane 3][k-half 2][rows_padded] of 16-byte units (eight consecutive k of one row):
 *   ixtts_gemm_x6_pack   splits a weight matrix W [N][K] (fp32, row-major, K a multiple of 64) once, at load, into
 *                        ixtts_gemm_x6_packed_bytes(N, K) bytes of device memory owned by the caller;
 *   ixtts_gemm_x6_split  splits activations A [rows][K] (row stride lda floats, 16-byte aligned rows) into planes of
 *                        ixtts_gemm_x6_rows_padded(rows) rows: (K/16) * 6 * rows_padded * 16 bytes owned by the caller;
 *   ixtts_gemm_x6_f32    C[M][N] (row stride ldc) (+)= A[row0 .. row0 + M) . W^T + bias over planes holding rows_total rows (a
 *                        row-shifted window of one split buffer = one tap of a k-tap Conv1d); bias_dev [N] or NULL; accumulate != 0
 *                        adds to what C holds (torch's addmm_); tile: 0 = chosen by the fill of the 256 CUs, 2 / 3 = 256x128 /
 *                        128x128 (A/B timing). */
size_t ixtts_gemm_x6_packed_bytes(int N, int K);
int ixtts_gemm_x6_pack(const float* w_dev, void* packed_dev, int N, int K, void* stream);
long ixtts_gemm_x6_rows_padded(long rows);
int ixtts_gemm_x6_split(const float* a_dev, long lda, void* planes_dev, long rows, int K, void* stream);
int ixtts_gemm_x6_f32(const void* a_planes_dev, long rows_total, long row0, const void* packed_dev, const float* bias_dev, float* c_dev, long ldc,
                      int M, int N, int K, int accumulate, int tile, void* stream);

/* ------------------------------------------------------------------------------------
 * Seam 2 -- BigVGAN-v2 generator
 * replaces: `BigVGAN.__init__/remove_weight_norm/forward`
 *   indextts/s2mel/modules/bigvgan/bigvgan.py:266-400 ; call site indextts/infer_v2.py:154-158,735
 */
#define IXTTS_BIGVGAN_MAX_STAGES 8
#define IXTTS_BIGVGAN_MAX_RESK 4

typedef struct ixtts_bigvgan_cfg {
  int num_mels;                 /* 80   (config.json:44) */
  int upsample_initial_channel; /* 1536 (config.json:13) */
  int n_stages;                 /* 6 */
  int upsample_rates[IXTTS_BIGVGAN_MAX_STAGES];        /* 4,4,2,2,2,2 */
  int upsample_kernel_sizes[IXTTS_BIGVGAN_MAX_STAGES]; /* 8,8,4,4,4,4 */
  int n_resblock_kernels;       /* 3 */
  int resblock_kernel_sizes[IXTTS_BIGVGAN_MAX_RESK];   /* 3,7,11 */
  int resblock_dilations[IXTTS_BIGVGAN_MAX_RESK][3];   /* 1,3,5 each */
  int max_frames;               /* workspace is sized for mel lengths up to this */
  int fast_sin;                 /* 0: libm-accurate sinf in the Snake (parity mode); 1: v_sin_f32 */
} ixtts_bigvgan_cfg;

typedef struct ixtts_bigvgan ixtts_bigvgan;

int ixtts_bigvgan_create(ixtts_bigvgan** out, const ixtts_bigvgan_cfg* cfg);
/* Upload one tensor of the weight-norm-FOLDED generator state dict by its reference name
 * ("conv_pre.weight", "ups.0.0.weight", "resblocks.3.convs1.2.bias",
 *  "resblocks.3.activations.5.act.alpha", "activation_post.act.beta", "conv_post.weight"...).
 * `data_host` is fp32 in the reference's own layout; the library re-packs it. */
int ixtts_bigvgan_set_tensor(ixtts_bigvgan* h, const char* name, const float* data_host, const int64_t* shape, int ndim);
/* Verifies every tensor was supplied. */
int ixtts_bigvgan_finalize(ixtts_bigvgan* h);
/* Packed weight arena (device), for a one-shot RCCL broadcast at load: after
 * `finalize` on rank 0 and `adopt_arena` elsewhere the handles are equivalent. */
int ixtts_bigvgan_arena(ixtts_bigvgan* h, void** ptr_dev, size_t* bytes);
int ixtts_bigvgan_adopt_arena(ixtts_bigvgan* h);
/* mel [B,num_mels,F] fp32 -> wav [B,1,F*prod(rates)] fp32, clamp(-1,1) (bigvgan.py:360-386). */
int ixtts_bigvgan_forward(ixtts_bigvgan* h, const float* mel_dev, int B, int F, float* wav_dev, void* stream);
/* Algorithmic conv FLOPs of one forward at F frames. */
double ixtts_bigvgan_flops(const ixtts_bigvgan* h, int B, int F);
int ixtts_bigvgan_destroy(ixtts_bigvgan* h);

/* ------------------------------------------------------------------------------------
 * Seam 1 -- autoregressive GPT-2 decode engine
 * replaces: the object DeepSpeed swaps in for `UnifiedVoice.inference_model`
 *   indextts/gpt/model_v2.py:433-446 (seam), :45-212 (GPT2InferenceModel),
 *   :663-734 (inference_speech -> store_mel_emb + generate), :554-596 (latent forward);
 *   trunk arithmetic indextts/gpt/transformers_gpt2.py:480-667,985-1184;
 *   token selection indextts/gpt/transformers_generation_utils.py:843-1070,3123-3297.
 */
#define IXTTS_DTYPE_F32 0
#define IXTTS_DTYPE_BF16 1

typedef struct ixtts_gpt_cfg {
  int model_dim;       /* 1280 */
  int layers;          /* 24   */
  int heads;           /* 20   (head dim must be 64) */
  int n_mel_codes;     /* 8194 (vocabulary of mel_head / mel_embedding) */
  int n_mel_pos;       /* rows of mel_pos_embedding (max_mel_tokens + 3 = 1818) */
  int n_text_tokens;   /* rows of text_embedding (12001) */
  int n_text_pos;      /* rows of text_pos_embedding (602) */
  int start_mel_token; /* 8192 */
  int stop_mel_token;  /* 8193 */
  int max_seq;         /* KV-cache capacity per sequence (prompt + generated) */
  int max_batch;       /* concurrent sequences (1 greedy; beams / segment batching later) */
  int weight_dtype;    /* IXTTS_DTYPE_F32 (parity mode) | IXTTS_DTYPE_BF16 (throughput mode) */
} ixtts_gpt_cfg;

typedef struct ixtts_sampler_cfg {
  float repetition_penalty; /* 10.0 (infer_v2.py:605); 1.0 disables                         */
  float temperature;        /* 0.8; ignored when do_sample == 0                              */
  int top_k;                /* 30; 0 (or >= vocabulary) disables the filter; any value for sampling, 1..128 in beam mode;
                               "greedy" of BASELINE configs == do_sample 0 or top_k 1 */
  float top_p;              /* 0.8                                                           */
  int do_sample;            /* 0: argmax of penalised logits; 1: multinomial after warpers.  Beam mode: 1 = beam-sample (the served
                               default), 0 = beam search proper -- the joint top 2 * num_beams, the warpers do not run
                               (transformers_generation_utils.py:1020,3520-3524) */
  int suppress_stop;        /* bench-only fixed-length mode: stop token forced to -inf       */
  uint64_t seed;            /* Philox seed for do_sample (cannot match torch's CPU stream)   */
  float typical_mass;       /* > 0: the custom TypicalLogitsWarper of `inference_speech(typical_sampling=True, typical_mass=...)`
                               (model_v2.py:717-722, utils/typical_sampling.py) after the repetition penalty; 0: off */
  float length_penalty;     /* beam mode: hypothesis score = sum_logprobs / generated_len ** length_penalty
                               (transformers_beam_search.py:947-951,989-1006); 0.0 is the served default (infer_v2.py:603) */
} ixtts_sampler_cfg;

typedef struct ixtts_gpt ixtts_gpt;

int ixtts_gpt_create(ixtts_gpt** out, const ixtts_gpt_cfg* cfg);
/* Upload one tensor of `UnifiedVoice.state_dict()` by name, fp32 host data in the
 * reference layout: gpt.h.{i}.{ln_1,ln_2}.{weight,bias}, gpt.h.{i}.attn.{c_attn,c_proj}.{weight,bias},
 * gpt.h.{i}.mlp.{c_fc,c_proj}.{weight,bias}, gpt.ln_f.*, final_norm.*, mel_head.*,
 * mel_embedding.weight, mel_pos_embedding.emb.weight.  Other names return IXTTS_ERR_NAME. */
int ixtts_gpt_set_tensor(ixtts_gpt* h, const char* name, const float* data_host, const int64_t* shape, int ndim);
int ixtts_gpt_finalize(ixtts_gpt* h);
int ixtts_gpt_arena(ixtts_gpt* h, void** ptr_dev, size_t* bytes);
int ixtts_gpt_adopt_arena(ixtts_gpt* h);
/* A second engine over the SAME device weights (one worker holds a register engine for single sequences / one beam group and a
 * wide engine for several beam groups; the reference has one model object, model_v2.py:433-446): `h` gives up its own arena and
 * reads `owner`'s.  Same model shape and weight type; `owner` must be finalized and outlive `h`. */
int ixtts_gpt_share_arena(ixtts_gpt* h, ixtts_gpt* owner);

/* `store_mel_emb(embeds)` + the prefill forward of generate() for sequence slot `b`
 * (model_v2.py:87-88,144-155): embeds_dev [P-1, D] fp32 = [pad][conds 34][text L+2] rows,
 * attention mask = zeros for the first `n_left_pad` rows then ones (model_v2.py:635-642).
 * Appends the start_mel_token row (mel_embedding[start] + mel_pos[0]) itself, fills the KV
 * cache, resets the slot's history to the fake ids [1]*(P-1)+[start] (model_v2.py:652-661)
 * and leaves the first-step logits ready. */
int ixtts_gpt_prefill(ixtts_gpt* h, int b, const float* embeds_dev, int n_rows, int n_left_pad, void* stream);
/* Run up to `n_steps` decode steps for slots [0, n_active): logits -> processors -> token ->
 * embed -> 24 layers.  No host synchronisation inside; a slot that emits stop_mel_token
 * keeps emitting it (generation_utils.py:3255-3256).  Tokens accumulate on the device. */
int ixtts_gpt_decode(ixtts_gpt* h, int n_active, int n_steps, const ixtts_sampler_cfg* sc, void* stream);
/* Synchronises `stream`; returns generated ids of slot b (up to and including the first
 * stop token) and whether the slot has finished. */
int ixtts_gpt_read(ixtts_gpt* h, int b, int32_t* ids_host, int cap, int* n_ids, int* finished, void* stream);
/* Raw fp32 logits of the most recent forward for slot b ([n_mel_codes], device -> host, sync). */
int ixtts_gpt_read_logits(ixtts_gpt* h, int b, float* logits_host, void* stream);
/* Probability vector [n_mel_codes] the last do_sample step drew from for slot b (after penalty,
 * temperature, top-k, top-p, softmax; zeros for removed ids).  Parity-test hook. */
int ixtts_gpt_read_probs(ixtts_gpt* h, int b, float* probs_host, void* stream);
/* Teacher forcing for parity tests: overrides the NEXT token chosen for slot b. */
int ixtts_gpt_force_next(ixtts_gpt* h, int b, int32_t token, void* stream);

/* Beam-sample, the reference's served default (`num_beams=3, do_sample=True`: infer_v2.py:598-605) --
 * `_beam_search` + `BeamSearchScorer` (transformers_generation_utils.py:3406-3565, transformers_beam_search.py:215-417,
 * 930-1013) with the whole per-step bookkeeping on the device.  Usage: ixtts_gpt_prefill(h, 0, ...) ->
 * ixtts_gpt_beam_begin(h, num_beams) (beams take slots 0..num_beams-1, needs max_batch >= num_beams) ->
 * ixtts_gpt_beam_decode(h, n_steps, cfg) as often as needed -> ixtts_gpt_beam_read. */
int ixtts_gpt_beam_begin(ixtts_gpt* h, int num_beams, void* stream);
int ixtts_gpt_beam_decode(ixtts_gpt* h, int n_steps, const ixtts_sampler_cfg* sc, void* stream);
/* Synchronises; runs `finalize` (best hypothesis, + eos if it fits in max_new) and returns it.  `done` = the scorer's
 * is_done flag.  Optional outputs (may be NULL): best score, and for tests the open beams' scores [num_beams], their
 * last tokens and the beam_idx of the last step. */
int ixtts_gpt_beam_read(ixtts_gpt* h, int max_new, int32_t* ids_host, int cap, int* n_ids, int* done, float* score,
                        float* beam_scores_host, int32_t* last_tokens_host, int32_t* beam_idx_host, void* stream);
/* Parity-test hook: the NEXT beam step uses these 2*num_beams flat draws (beam*V + token) instead of sampling. */
int ixtts_gpt_beam_force(ixtts_gpt* h, const int32_t* picks_host, int n, void* stream);

/* Beam GROUPS: the reference runs `inference_speech` once per text segment, one after another (infer_v2.py:616-658), each a
 * `_beam_search` over num_beams sequences.  Here the segments of a request (or of several requests) decode TOGETHER: group g
 * owns slots g*num_beams .. g*num_beams+num_beams-1 and its own scorer state (beam scores, hypotheses, done flag, forced
 * draws), the weights are read once per step for all groups.  An engine of up to 4 slots holds one group (the calls above ==
 * group 0); a wide engine (max_batch 5..16, bf16) holds floor(max_batch / num_beams) of them.  A group's tokens do not depend
 * on which other groups step with it.  Usage per segment: ixtts_gpt_prefill(h, g*num_beams, ...) ->
 * ixtts_gpt_beam_begin_group(h, g, num_beams, rng_stream) (rng_stream selects the group's random stream: 0 is the stream
 * ixtts_gpt_beam_begin uses, so segment i decoded with rng_stream i draws the same numbers in any group) -> repeated
 * ixtts_gpt_beam_decode_groups(h, n_groups, n_steps, cfg) stepping groups 0..n_groups-1 (finished or parked groups inside
 * that range are no-ops) -> ixtts_gpt_beam_read_group.  ixtts_gpt_beam_park_group takes a group out of the stepping set
 * until its next begin (its slots keep running through the layers with their last token; nothing is read from them). */
int ixtts_gpt_beam_begin_group(ixtts_gpt* h, int group, int num_beams, uint64_t rng_stream, void* stream);
int ixtts_gpt_beam_park_group(ixtts_gpt* h, int group, void* stream);
int ixtts_gpt_beam_decode_groups(ixtts_gpt* h, int n_groups, int n_steps, const ixtts_sampler_cfg* sc, void* stream);
int ixtts_gpt_beam_read_group(ixtts_gpt* h, int group, int max_new, int32_t* ids_host, int cap, int* n_ids, int* done, float* score,
                              float* beam_scores_host, int32_t* last_tokens_host, int32_t* beam_idx_host, void* stream);
int ixtts_gpt_beam_force_group(ixtts_gpt* h, int group, const int32_t* picks_host, int n, void* stream);

/* `UnifiedVoice.forward(...)->get_logits(return_latent=True)` (model_v2.py:554-596,486-512):
 * prefix_dev [n_prefix, D] = [conds 34 ; text_emb L+2] rows; codes_dev [n] int32 mel codes.
 * Embeds [start, codes, stop] with mel positions 0..n+1, runs the full causal trunk,
 * ln_f + final_norm, writes the first n mel rows to latent_dev [n, D] fp32. */
int ixtts_gpt_latent(ixtts_gpt* h, const float* prefix_dev, int n_prefix, const int32_t* codes_dev, int n,
                     float* latent_dev, void* stream);

/* Microbench hook for bench.py's roofline leg: launches the decode-step GEMV kernel of
 * `which` (0 qkv, 1 attn-out, 2 fc, 3 mlp-out, 4 head) for layer `layer` once on `stream`. */
int ixtts_gpt_bench_gemv(ixtts_gpt* h, int which, int layer, int batch, void* stream);
/* Algorithmic HBM bytes of one decode step at batch B and context S. */
double ixtts_gpt_step_bytes(const ixtts_gpt* h, int B, int S);

/* Largest `max_batch` (decode slots decoded together, weights read once per step for all of them) this build supports. */
int ixtts_gpt_max_batch(void);

int ixtts_gpt_destroy(ixtts_gpt* h);

#ifdef __cplusplus
}
#endif
#endif /* IXTTS_HIP_H */
