"""CPU restatement of the IndexTTS2 autoregressive GPT path (fp32).  TEST INFRASTRUCTURE ONLY.

Rows G0-G9 of SURVEY.md section 8(a).  Reference files (relative to the reference root):
  indextts/gpt/model_v2.py:90-212      GPT2InferenceModel (prepare_inputs / forward / reorder)
  indextts/gpt/model_v2.py:554-661     UnifiedVoice.forward (latent pass), prepare_gpt_inputs
  indextts/gpt/model_v2.py:663-734     inference_speech
  indextts/gpt/transformers_gpt2.py:129-348,464-568,571-667,985-1184   GPT-2 trunk (text twin
      of third-party transformers==4.52.1, pyproject.toml:58)
  indextts/gpt/transformers_generation_utils.py:843-1070,3123-3297      processors, _sample

Weights are a flat dict keyed like `UnifiedVoice.state_dict()`:
  gpt.h.{i}.ln_1.{weight,bias}  gpt.h.{i}.attn.c_attn.{weight[D,3D],bias}
  gpt.h.{i}.attn.c_proj.{weight[D,D],bias}  gpt.h.{i}.ln_2.*  gpt.h.{i}.mlp.c_fc.{weight[D,4D],bias}
  gpt.h.{i}.mlp.c_proj.{weight[4D,D],bias}  gpt.ln_f.*  final_norm.*  mel_head.{weight[V,D],bias}
  mel_embedding.weight  mel_pos_embedding.emb.weight  text_embedding.weight
  text_pos_embedding.emb.weight  speed_emb.weight
"""
import math

import torch
import torch.nn.functional as F

NEG = torch.finfo(torch.float32).min


def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm(eps=1e-5) (transformers_gpt2.py:598-600,1164; model_v2.py:398)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def gelu_new(x):
    """ACT2FN['gelu_new'] (config default activation_function; transformers_gpt2.py:571-585)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


class GptOracle:
    def __init__(self, W, n_layer, n_head):
        self.W = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in W.items() if k.startswith(("gpt.", "final_norm", "mel_", "text_", "speed_emb"))}
        self.L = n_layer
        self.H = n_head
        self.D = self.W["gpt.ln_f.weight"].shape[0]
        self.dh = self.D // n_head

    # ------------------------------------------------------------------ trunk
    def trunk(self, emb, past=None, key_mask=None):
        """GPT2Model.forward with inputs_embeds, wpe == 0 (model_v2.py:22-23,272-274).

        emb [B,T,D]; past: list of (k,v) each [B,H,S,dh] or None; key_mask [B,S+T] of 0/1
        (HF attention_mask; None == all ones).  Returns (ln_f(h) [B,T,D], new past).

        Attention = softmax(q k^T / sqrt(dh) + causal + padding) v with the additive
        finfo.min mask of _prepare_4d_causal_attention_mask_for_sdpa
        (transformers_gpt2.py:1045-1051, 531-558).
        """
        W, H, dh = self.W, self.H, self.dh
        B, T, D = emb.shape
        S = 0 if past is None else past[0][0].shape[2]
        # query row i (absolute position S+i) may see key j iff j <= S+i and key_mask[j]==1
        qpos = torch.arange(S, S + T).view(T, 1)
        kpos = torch.arange(S + T).view(1, S + T)
        allow = (kpos <= qpos).view(1, 1, T, S + T)
        if key_mask is not None:
            allow = allow & (torch.as_tensor(key_mask).view(B, 1, 1, S + T) != 0)
        add_mask = torch.zeros(allow.shape, dtype=torch.float32).masked_fill(~allow, NEG)
        h = emb
        new_past = []
        for i in range(self.L):
            p = f"gpt.h.{i}."
            a = layer_norm(h, W[p + "ln_1.weight"], W[p + "ln_1.bias"])
            qkv = a @ W[p + "attn.c_attn.weight"] + W[p + "attn.c_attn.bias"]  # Conv1D: x@W+b
            q, k, v = qkv.split(D, dim=-1)
            q = q.view(B, T, H, dh).transpose(1, 2)
            k = k.view(B, T, H, dh).transpose(1, 2)
            v = v.view(B, T, H, dh).transpose(1, 2)
            if past is not None:
                k = torch.cat((past[i][0], k), dim=2)
                v = torch.cat((past[i][1], v), dim=2)
            new_past.append((k, v))
            sc = (q @ k.transpose(-1, -2)) / math.sqrt(dh) + add_mask
            att = torch.softmax(sc, dim=-1) @ v
            att = att.transpose(1, 2).reshape(B, T, D)
            h = h + (att @ W[p + "attn.c_proj.weight"] + W[p + "attn.c_proj.bias"])
            m = layer_norm(h, W[p + "ln_2.weight"], W[p + "ln_2.bias"])
            m = gelu_new(m @ W[p + "mlp.c_fc.weight"] + W[p + "mlp.c_fc.bias"])
            h = h + (m @ W[p + "mlp.c_proj.weight"] + W[p + "mlp.c_proj.bias"])
        h = layer_norm(h, W["gpt.ln_f.weight"], W["gpt.ln_f.bias"])
        return h, new_past

    def head(self, h):
        """lm_head = Sequential(final_norm, mel_head) (model_v2.py:53,185) -> fp32 logits."""
        W = self.W
        x = layer_norm(h, W["final_norm.weight"], W["final_norm.bias"])
        return x @ W["mel_head.weight"].t() + W["mel_head.bias"]

    # ---------------------------------------------------------- prompt (G0)
    def prepare_gpt_inputs(self, conds_latent, text_ids, start_text=0, stop_text=1, start_mel=8192):
        """UnifiedVoice.prepare_gpt_inputs (model_v2.py:598-661), batch 1.

        conds_latent [34,D] (cond+emo, speed rows); text_ids int [L].
        Returns fake_ids [P] int64, embeds [P-1,D], mask [P] int64.
        """
        W = self.W
        text_ids = torch.as_tensor(text_ids, dtype=torch.long)
        L = text_ids.numel()
        valid = (text_ids != stop_text) & (text_ids != start_text)
        t = text_ids[valid]
        t = torch.cat((torch.tensor([start_text]), t, torch.tensor([stop_text])))
        temb = W["text_embedding.weight"][t] + W["text_pos_embedding.emb.weight"][: t.numel()]
        target_len = conds_latent.shape[0] + L + 2
        mask = torch.ones(target_len + 1, dtype=torch.long)
        pad = L + 2 - t.numel()
        parts = [torch.as_tensor(conds_latent, dtype=torch.float32), temb]
        if pad > 0:
            parts.insert(0, torch.zeros(pad, self.D))
            mask[:pad] = 0
        embeds = torch.cat(parts, dim=0)
        fake = torch.ones(target_len + 1, dtype=torch.long)
        fake[-1] = start_mel
        return fake, embeds, mask

    def conds_latent(self, cond32, emo_vec):
        """inference_speech (model_v2.py:693-696): cat(cond + emo_vec, speed_emb[1], speed_emb[0])."""
        W = self.W
        c = torch.as_tensor(cond32, dtype=torch.float32) + torch.as_tensor(emo_vec, dtype=torch.float32).view(1, -1)
        return torch.cat((c, W["speed_emb.weight"][1:2], W["speed_emb.weight"][0:1]), dim=0)

    # ------------------------------------------------------- prefill / decode
    def prefill(self, embeds, mask, start_mel=8192):
        """GPT2InferenceModel.forward, input_ids.shape[1] != 1 (model_v2.py:144-155).

        embeds [P-1,D] (the stored `cached_mel_emb`), mask [P].  The start_mel_token row is
        mel_embedding[8192] + mel_pos_embedding[0].  Returns (logits [V] of the last row, past).
        """
        W = self.W
        row = W["mel_embedding.weight"][start_mel] + W["mel_pos_embedding.emb.weight"][0]
        emb = torch.cat((embeds, row.view(1, -1)), dim=0).unsqueeze(0)
        h, past = self.trunk(emb, None, torch.as_tensor(mask).view(1, -1))
        return self.head(h[:, -1])[0], past

    def decode_step(self, token, k, past, mask_prefix):
        """GPT2InferenceModel.forward, input_ids.shape[1] == 1 (model_v2.py:156-160).

        `token` is the k-th generated id (k >= 1); its position row is
        mel_pos_embedding[attention_mask.shape[1] - mel_len] = [k + 1]   (SURVEY.md F6).
        """
        W = self.W
        emb = (W["mel_embedding.weight"][int(token)] + W["mel_pos_embedding.emb.weight"][k + 1]).view(1, 1, -1)
        S = past[0][0].shape[2]
        km = torch.cat((torch.as_tensor(mask_prefix).view(1, -1), torch.ones(1, S + 1 - len(mask_prefix), dtype=torch.long)), dim=1)
        h, past = self.trunk(emb, past, km)
        return self.head(h[:, -1])[0], past

    def decode_step_batch(self, tokens, k, past, mask_prefix):
        """`decode_step` for B sequences that share the prompt mask and sit at the same step k (the beams of one
        `_beam_search`, model_v2.py:156-160 with input_ids [B, 1]); past: list of (k, v) each [B,H,S,dh].
        Returns (logits [B,V], past)."""
        W = self.W
        tok = torch.as_tensor(tokens, dtype=torch.long)
        emb = (W["mel_embedding.weight"][tok] + W["mel_pos_embedding.emb.weight"][k + 1]).unsqueeze(1)
        S = past[0][0].shape[2]
        km = torch.cat((torch.as_tensor(mask_prefix).view(1, -1), torch.ones(1, S + 1 - len(mask_prefix), dtype=torch.long)), dim=1)
        h, past = self.trunk(emb, past, km.expand(tok.numel(), -1))
        return self.head(h[:, -1]), past

    # -------------------------------------------------------- latent pass (G9)
    def latent_pass(self, conds_latent, text_ids, codes, start_text=0, stop_text=1, start_mel=8192, stop_mel=8193):
        """UnifiedVoice.forward(...)->get_logits(return_latent=True) (model_v2.py:554-596,486-512).

        conds_latent [34,D]; text_ids [L]; codes [n].  Returns latent [n, D].
        """
        W = self.W
        text_ids = torch.as_tensor(text_ids, dtype=torch.long)
        codes = torch.as_tensor(codes, dtype=torch.long)
        t = torch.cat((torch.tensor([start_text]), text_ids, torch.tensor([stop_text])))
        temb = W["text_embedding.weight"][t] + W["text_pos_embedding.emb.weight"][: t.numel()]
        m = torch.cat((torch.tensor([start_mel]), codes, torch.tensor([stop_mel])))
        memb = W["mel_embedding.weight"][m] + W["mel_pos_embedding.emb.weight"][: m.numel()]
        conds = torch.as_tensor(conds_latent, dtype=torch.float32)
        emb = torch.cat((conds, temb, memb), dim=0).unsqueeze(0)
        h, _ = self.trunk(emb, None, None)
        enc = layer_norm(h[:, conds.shape[0]:], W["final_norm.weight"], W["final_norm.bias"])
        mel = enc[0, -m.numel():]
        return mel[:-2]


# ------------------------------------------------------------------- sampler (G8)
def repetition_penalty(scores, history, theta):
    """RepetitionPenaltyLogitsProcessor: s<0 ? s*theta : s/theta on ids in history."""
    scores = scores.clone()
    idx = torch.unique(torch.as_tensor(history, dtype=torch.long))
    s = scores[idx]
    scores[idx] = torch.where(s < 0, s * theta, s / theta)
    return scores


def top_k_filter(scores, k, min_keep=1):
    k = min(max(k, min_keep), scores.numel())
    kth = torch.topk(scores, k).values[-1]
    return scores.masked_fill(scores < kth, float("-inf"))


def top_p_filter(scores, top_p, min_keep=1):
    """TopPLogitsWarper: sort ascending, drop the prefix whose cumulative prob <= 1 - top_p."""
    sorted_logits, sorted_idx = torch.sort(scores, descending=False)
    cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    remove_sorted = cum <= (1 - top_p)
    remove_sorted[-min_keep:] = False
    remove = torch.zeros_like(remove_sorted).scatter(0, sorted_idx, remove_sorted)
    return scores.masked_fill(remove, float("-inf"))


def typical_filter(scores, mass=0.9, min_keep=1):
    """indextts/utils/typical_sampling.py:8-30 (the custom processor `inference_speech(typical_sampling=True)` appends,
    model_v2.py:717-722): keep the tokens whose surprise -log p is closest to the entropy until their mass reaches `mass`."""
    normalized = torch.log_softmax(scores, dim=-1)
    p = torch.exp(normalized)
    ent = -(normalized * p).nansum(-1, keepdim=True)
    shifted = torch.abs((-normalized) - ent)
    sorted_scores, sorted_idx = torch.sort(shifted, descending=False)
    cum = scores.gather(-1, sorted_idx).softmax(dim=-1).cumsum(dim=-1)
    last = int((cum < mass).sum())
    remove_sorted = sorted_scores > sorted_scores[min(last, sorted_scores.numel() - 1)]
    if min_keep > 1:
        remove_sorted[:min_keep] = False
    remove = torch.zeros_like(remove_sorted).scatter(0, sorted_idx, remove_sorted)
    return scores.masked_fill(remove, float("-inf"))


def process_logits(logits, history, theta=10.0, temperature=None, top_k=None, top_p=None, min_keep=1, suppress=None, typical_mass=None):
    """Processor chain in `_get_logits_processor` order (generation_utils.py:900-901,1020-1044); a custom processor
    (typical sampling) sits between the repetition penalty and the warpers.

    `suppress` (bench-only fixed-length mode): ids forced to -inf first.
    """
    s = logits.to(torch.float32).clone()
    if suppress is not None:
        s[torch.as_tensor(suppress, dtype=torch.long)] = float("-inf")
    if theta is not None and theta != 1.0:
        s = repetition_penalty(s, history, theta)
    if typical_mass is not None:
        s = typical_filter(s, typical_mass, min_keep)
    if temperature is not None and temperature != 1.0:
        s = s / temperature
    if top_k is not None and top_k > 0:
        s = top_k_filter(s, top_k, min_keep)
    if top_p is not None and top_p < 1.0:
        s = top_p_filter(s, top_p, min_keep)
    return s


def generate_greedy(oracle, embeds, mask, max_new, theta=10.0, stop_mel=8193, start_mel=8192,
                    suppress_stop=False, return_logits=False, forced=None):
    """`_sample` with do_sample=False / top_k=1 (generation_utils.py:3196-3269).

    History for the penalty = fake prefix [1]*(P-1)+[8192] + generated (model_v2.py:652-661, F7).
    Returns (ids list, per-step top-2 margins[, per-step raw logits]).
    `forced`: optional teacher-forcing ids (the argmax is still recorded in margins/logits).
    """
    P = len(mask)
    history = [1] * (P - 1) + [start_mel]
    logits, past = oracle.prefill(embeds, mask, start_mel)
    ids, margins, all_logits = [], [], []
    for k in range(1, max_new + 1):
        s = process_logits(logits, history, theta, suppress=[stop_mel] if suppress_stop else None)
        top2 = torch.topk(s, 2).values
        margins.append(float(top2[0] - top2[1]))
        if return_logits:
            all_logits.append(logits.clone())
        tok = int(torch.argmax(s))
        if forced is not None:
            tok = int(forced[k - 1])
        ids.append(tok)
        history.append(tok)
        if tok == stop_mel or k == max_new:
            break
        logits, past = oracle.decode_step(tok, k, past, mask)
    if return_logits:
        return ids, margins, torch.stack(all_logits)
    return ids, margins


def teacher_forced_logits(oracle, embeds, mask, ids, start_mel=8192):
    """The decode loop's logits for a GIVEN id sequence in ONE causal pass (what `generate_greedy(forced=ids)` computes
    step by step through the KV cache; tests/test_oracle_golden.py holds the two to each other): rows = the stored prompt
    embeds, the start_mel row at mel position 0, then the k-th id (k >= 1) at mel position k + 1 (SURVEY F6,
    model_v2.py:144-160); left padding through the key mask.  Returns logits [len(ids) + 1, V]: row k is the
    distribution token k + 1 is chosen from (row 0 = the prefill logits)."""
    W = oracle.W
    ids = torch.as_tensor(ids, dtype=torch.long)
    n = ids.numel()
    rows = [torch.as_tensor(embeds, dtype=torch.float32), (W["mel_embedding.weight"][start_mel] + W["mel_pos_embedding.emb.weight"][0]).view(1, -1)]
    if n:
        rows.append(W["mel_embedding.weight"][ids] + W["mel_pos_embedding.emb.weight"][2: n + 2])
    emb = torch.cat(rows, dim=0).unsqueeze(0)
    P = len(mask)
    km = torch.cat((torch.as_tensor(mask).view(1, -1), torch.ones(1, n, dtype=torch.long)), dim=1)
    h, _ = oracle.trunk(emb, None, km)
    return oracle.head(h[0, P - 1:])


def greedy_choices(logits_rows, P, ids, theta=10.0, stop_mel=8193, start_mel=8192, suppress_stop=False):
    """What `_sample` with top_k=1 picks at every step of a teacher-forced run: the argmax of the repetition-penalised row k
    given the history [1]*(P-1)+[8192]+ids[:k] (F7), and its top-2 margin.  Returns (choices [n], margins [n])."""
    hist = [1] * (P - 1) + [start_mel]
    out, margins = [], []
    for k in range(len(ids)):
        s = process_logits(logits_rows[k], hist, theta, suppress=[stop_mel] if suppress_stop else None)
        top2 = torch.topk(s, 2)
        out.append(int(top2.indices[0]))
        margins.append(float(top2.values[0] - top2.values[1]))
        hist.append(int(ids[k]))
    return out, margins


# ------------------------------------------------------------------- beam-sample (G8, served default)
class BeamHyps:
    """BeamHypotheses (indextts/gpt/transformers_beam_search.py:930-1013), length_penalty / early_stopping=False."""

    def __init__(self, num_beams, length_penalty=0.0):
        self.num_beams = num_beams
        self.length_penalty = length_penalty
        self.beams = []  # (score, tokens)
        self.worst_score = 1e9

    def add(self, hyp, sum_logprobs, generated_len):
        score = sum_logprobs / (generated_len ** self.length_penalty)
        if len(self.beams) < self.num_beams or score > self.worst_score:
            self.beams.append((score, list(hyp)))
            if len(self.beams) > self.num_beams:
                order = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[order[0][1]]
                self.worst_score = order[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len, prompt_len):
        if len(self.beams) < self.num_beams:
            return False
        highest = best_sum_logprobs / (cur_len - prompt_len) ** self.length_penalty
        return self.worst_score >= highest


def beam_process(hyps, done, histories, next_scores, next_tokens, next_indices, eos, prompt_len):
    """BeamSearchScorer.process for one batch item (transformers_beam_search.py:215-318).

    histories: list of per-beam generated-token lists; candidates sorted by score descending.
    Returns (next_beam_scores, next_beam_tokens, next_beam_indices, done).
    """
    nb = hyps.num_beams
    cur_len = prompt_len + len(histories[0]) + 1
    if done:
        return [0.0] * nb, [eos] * nb, [0] * nb, True
    out_s, out_t, out_i = [], [], []
    for rank, (tok, sc, bi) in enumerate(zip(next_tokens, next_scores, next_indices)):
        if tok == eos:
            if rank >= nb:
                continue
            hyps.add(histories[bi], sc, generated_len=cur_len - prompt_len)
        else:
            out_s.append(sc)
            out_t.append(tok)
            out_i.append(bi)
        if len(out_s) == nb:
            break
    assert len(out_s) == nb, "fewer than num_beams non-eos candidates"
    done = done or hyps.is_done(max(next_scores), cur_len, prompt_len)
    return out_s, out_t, out_i, done


def beam_finalize(hyps, done, histories, beam_scores, eos, max_new):
    """BeamSearchScorer.finalize, num_return_sequences = 1 (transformers_beam_search.py:320-417)."""
    if not done:
        for b in range(hyps.num_beams):
            hyps.add(histories[b], beam_scores[b], generated_len=len(histories[b]))
    best = sorted(hyps.beams, key=lambda x: x[0])[-1]
    seq = list(best[1])
    if len(seq) < max_new:
        seq.append(eos)
    return seq, best[0]


def beam_scores_step(logits_rows, histories_full, beam_scores, theta, temperature, top_k, top_p, suppress=None):
    """log_softmax -> processors (min_tokens_to_keep = 2) -> + beam score (generation_utils.py:3473-3481).
    `suppress` (bench-only fixed-length mode): those ids' log-probabilities forced to -inf before the processors."""
    rows = []
    for b in range(len(beam_scores)):
        lp = torch.log_softmax(logits_rows[b].to(torch.float32), dim=-1)
        s = process_logits(lp, histories_full[b], theta, temperature, top_k, top_p, min_keep=2, suppress=suppress)
        rows.append(s + beam_scores[b])
    return torch.stack(rows)  # [num_beams, V]


def generate_beam_sample(oracle, embeds, mask, max_new, num_beams=3, theta=10.0, temperature=0.8, top_k=30, top_p=0.8,
                         stop_mel=8193, start_mel=8192, sampler=None, generator=None, trace=None, length_penalty=0.0,
                         batched=False, suppress_stop=False, keep_logits=None):
    """`_beam_search` with do_sample=True (generation_utils.py:3406-3565): the served default.

    `sampler(scores_flat[num_beams*V]) -> 2*num_beams flat indices` lets a test force the draws;
    default = softmax + torch.multinomial without replacement.
    `batched`: the beams step through the trunk as one batch (as the reference does: input_ids [num_beams, 1], the cache
    reordered by index_select, model_v2.py:199-212) instead of one by one -- the production-width tests use it, the weights
    being read once per step; tests/test_oracle_golden.py holds the two forms to each other.
    `keep_logits`: steps s whose per-beam logits (the distributions step s + 1 draws from) go into trace[s - 1]["logits"].
    """
    P = len(mask)
    prefix = [1] * (P - 1) + [start_mel]
    logits0, past0 = oracle.prefill(embeds, mask, start_mel)
    V = logits0.numel()
    logits = [logits0.clone() for _ in range(num_beams)]
    pasts = [past0 for _ in range(num_beams)]
    if batched:
        pasts = [(k.expand(num_beams, -1, -1, -1), v.expand(num_beams, -1, -1, -1)) for k, v in past0]
    hist = [[] for _ in range(num_beams)]
    beam_scores = [0.0] + [-1e9] * (num_beams - 1)
    hyps, done = BeamHyps(num_beams, length_penalty), False
    for step in range(1, max_new + 1):
        scores = beam_scores_step(logits, [prefix + h for h in hist], beam_scores, theta, temperature, top_k, top_p,
                                  suppress=[stop_mel] if suppress_stop else None)
        flat = scores.reshape(-1)
        if sampler is not None:
            picks = sampler(flat, step)
        else:
            picks = torch.multinomial(torch.softmax(flat, -1), 2 * num_beams, generator=generator)
        picks = torch.as_tensor(picks, dtype=torch.long)
        sc = flat[picks]
        sc, order = torch.sort(sc, descending=True)
        picks = picks[order]
        nidx = (picks // V).tolist()
        ntok = (picks % V).tolist()
        ns, nt, ni, done = beam_process(hyps, done, hist, sc.tolist(), ntok, nidx, stop_mel, P)
        if trace is not None:
            trace.append(dict(picks=picks.tolist(), scores=sc.tolist(), next_scores=list(ns), next_tokens=list(nt), next_indices=list(ni), done=done))
        hist = [hist[ni[j]] + [nt[j]] for j in range(num_beams)]
        beam_scores = list(ns)
        if done or step == max_new:
            break
        if batched:
            if list(ni) != list(range(num_beams)):  # _reorder_cache (model_v2.py:199-212); the identity permutation moves nothing
                idx = torch.as_tensor(ni, dtype=torch.long)
                pasts = [(k.index_select(0, idx), v.index_select(0, idx)) for k, v in pasts]
            lg, pasts = oracle.decode_step_batch(nt, step, pasts, mask)
            logits = [lg[j] for j in range(num_beams)]
        else:
            pasts = [pasts[ni[j]] for j in range(num_beams)]
            new_logits, new_pasts = [], []
            for j in range(num_beams):
                lg, pj = oracle.decode_step(nt[j], step, pasts[j], mask)
                new_logits.append(lg)
                new_pasts.append(pj)
            logits, pasts = new_logits, new_pasts
        if trace is not None and keep_logits is not None and step in keep_logits:
            trace[-1]["logits"] = torch.stack(list(logits)).clone()
    seq, score = beam_finalize(hyps, done, hist, beam_scores, stop_mel, max_new)
    return seq, score


def generate_beam_search(oracle, embeds, mask, max_new, num_beams=3, theta=10.0, **kw):
    """`_beam_search` with do_sample=False (generation_utils.py:3511-3524): the 2 * num_beams candidates are the TOP of the joint
    penalised log-probabilities + beam scores (`torch.topk`, sorted); the warpers (temperature / top-k / top-p) are sampling-only
    (generation_utils.py:1020) and do not run.  Everything else -- scorer, cache reorder, finalize -- is the loop above.
    Pinned by tests/golden/gpt_beam_search.npz (the reference's scorer + model forward, num_beams 2 / 3 / 4)."""
    return generate_beam_sample(oracle, embeds, mask, max_new, num_beams=num_beams, theta=theta, temperature=None, top_k=None, top_p=None,
                                sampler=lambda flat, step: torch.topk(flat, 2 * num_beams, largest=True, sorted=True).indices, **kw)


def beam_replay(oracle, embeds, mask, step_tokens, step_src, theta=10.0, temperature=0.8, top_k=30, top_p=0.8, stop_mel=8193,
                start_mel=8192, suppress_stop=False, keep_logits=()):
    """A GIVEN beam-sample run (per step: the token each new beam took and the beam it continues) pushed through the oracle:
    what `_beam_search` would have scored it.  Returns a list with, per step, `inc` [num_beams] = the processed log-probability
    of (source beam, token) -- the beam score's increment (-inf where the oracle's TopK / TopP removed that token), `kept`
    [num_beams] bools, `amp` [num_beams] = d(increment) / d(log-probability) of that pair through the processors (theta /
    temperature for a token of the history, whose negative log-probability the repetition penalty multiplies by theta; 1 /
    temperature otherwise) -- what a log-probability error is amplified by -- and `logits` [num_beams, V] after the steps in
    `keep_logits`.  The beams step as one batch."""
    P = len(mask)
    prefix = [1] * (P - 1) + [start_mel]
    nb = len(step_tokens[0])
    logits0, past0 = oracle.prefill(embeds, mask, start_mel)
    logits = [logits0.clone() for _ in range(nb)]
    pasts = [(k.expand(nb, -1, -1, -1), v.expand(nb, -1, -1, -1)) for k, v in past0]
    hist = [[] for _ in range(nb)]
    out = []
    for step, (toks, src) in enumerate(zip(step_tokens, step_src), start=1):
        toks, src = [int(t) for t in toks], [int(b) for b in src]
        scores = beam_scores_step(logits, [prefix + h for h in hist], [0.0] * nb, theta, temperature, top_k, top_p,
                                  suppress=[stop_mel] if suppress_stop else None)
        inc = [float(scores[src[j], toks[j]]) for j in range(nb)]
        t_ = temperature if temperature else 1.0
        amp = [((theta if theta else 1.0) if toks[j] in set(prefix + hist[src[j]]) else 1.0) / t_ for j in range(nb)]
        rec = dict(inc=inc, kept=[v != float("-inf") for v in inc], amp=amp)
        hist = [hist[src[j]] + [toks[j]] for j in range(nb)]
        if src != list(range(nb)):
            idx = torch.as_tensor(src, dtype=torch.long)
            pasts = [(k.index_select(0, idx), v.index_select(0, idx)) for k, v in pasts]
        lg, pasts = oracle.decode_step_batch(toks, step, pasts, mask)
        logits = [lg[j] for j in range(nb)]
        if step in keep_logits:
            rec["logits"] = lg.clone()
        out.append(rec)
    return out
