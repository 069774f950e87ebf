"""Generate tests/golden/*.npz from the REFERENCE's own module classes (build container only).

    python tests/golden/make_golden.py

Imports /root/reference through tests/golden/ref_harness.py, instantiates the reference
classes (Activation1d/SnakeBeta, BigVGAN, UnifiedVoice/GPT2InferenceModel and the
third-party HF logits processors its generate() assembles) with seeded random weights,
and stores inputs + reference outputs.  The .npz files are data only (no reference
source text).  tests/test_oracle_golden.py pins oracle/ against them; the GPU tests
compare the HIP path with both.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness  # noqa: E402

ref_harness.install()

import voice_tts_amd.weights as WR  # noqa: E402

torch.set_grad_enabled(False)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ----------------------------------------------------------------------------- K1 / V4
def gen_aa_snake():
    from indextts.s2mel.modules.bigvgan.activations import SnakeBeta
    from indextts.s2mel.modules.bigvgan.alias_free_activation.torch.act import Activation1d

    g = torch.Generator().manual_seed(11)
    out = {}
    for (C, T) in [(3, 1), (3, 2), (5, 5), (4, 11), (24, 64), (6, 300), (2, 4097)]:
        act = Activation1d(activation=SnakeBeta(C, alpha_logscale=True))
        act.act.alpha.data = torch.randn(C, generator=g) * 0.5
        act.act.beta.data = torch.randn(C, generator=g) * 0.5
        x = torch.randn(2, C, T, generator=g) * 1.5
        y = act(x)
        tag = f"c{C}_t{T}"
        out[f"x_{tag}"] = x
        out[f"la_{tag}"] = act.act.alpha.data
        out[f"lb_{tag}"] = act.act.beta.data
        out[f"y_{tag}"] = y
        out["up_filter"] = act.upsample.filter.flatten()
        out["down_filter"] = act.downsample.lowpass.filter.flatten()
    save("aa_snake.npz", **out)


# ----------------------------------------------------------------------------- V0-V5
def build_ref_bigvgan(cfg, W):
    from indextts.s2mel.modules.bigvgan import bigvgan as RB

    h = RB.load_hparams_from_json(os.path.join(ref_harness.REF, "indextts/s2mel/modules/bigvgan/config.json"))
    h["upsample_initial_channel"] = cfg["upsample_initial_channel"]
    m = RB.BigVGAN(h, use_cuda_kernel=False)
    m.remove_weight_norm()
    m.eval()
    sd = m.state_dict()
    new = {}
    for k in sd:
        if k in W:
            assert tuple(sd[k].shape) == tuple(W[k].shape), k
            new[k] = W[k]
        else:
            assert k.endswith("filter"), k  # registered filter buffers only
            new[k] = sd[k]
    missing = set(W) - set(sd)
    assert not missing, missing
    m.load_state_dict(new, strict=True)
    return m


def gen_bigvgan():
    cfg = WR.tiny_bigvgan_cfg(64)
    W = WR.make_bigvgan_weights(cfg, seed=21)
    m = build_ref_bigvgan(cfg, W)
    g = torch.Generator().manual_seed(22)
    out = {"seed": 21, "upsample_initial_channel": 64}
    for F in (1, 7, 40):
        mel = (torch.randn(1, 80, F, generator=g) * 2 - 4).clamp(-11.5, 2)
        out[f"mel_f{F}"] = mel
        out[f"wav_f{F}"] = m(mel)
    save("bigvgan_tiny.npz", **out)


# ----------------------------------------------------------------------------- G0-G9
TINY_COND = dict(output_size=32, linear_units=64, attention_heads=2, num_blocks=1, input_layer="conv2d2", perceiver_mult=2)


def build_ref_gpt(cfg, W):
    from indextts.gpt.model_v2 import UnifiedVoice

    uv = UnifiedVoice(
        layers=cfg["layers"], model_dim=cfg["model_dim"], heads=cfg["heads"],
        max_text_tokens=cfg["max_text_tokens"], max_mel_tokens=cfg["max_mel_tokens"],
        number_text_tokens=cfg["number_text_tokens"], number_mel_codes=cfg["number_mel_codes"],
        start_mel_token=cfg["start_mel_token"], stop_mel_token=cfg["stop_mel_token"],
        start_text_token=cfg["start_text_token"], stop_text_token=cfg["stop_text_token"],
        mel_length_compression=1024, use_mel_codes_as_input=True, train_solo_embeddings=False,
        condition_type="conformer_perceiver", condition_module=TINY_COND, emo_condition_module=TINY_COND,
    ).eval()
    sd = uv.state_dict()
    for k, v in W.items():
        assert k in sd and tuple(sd[k].shape) == tuple(v.shape), (k, tuple(v.shape), tuple(sd[k].shape) if k in sd else None)
    res = uv.load_state_dict(W, strict=False)
    assert not res.unexpected_keys
    uv.post_init_gpt2_config(use_deepspeed=False, kv_cache=True, half=False)
    return uv


def ref_greedy(uv, conds_latent, text_ids, n_steps, theta=10.0, forced=None):
    """Harness loop over the reference's own prepare_inputs_for_generation + forward."""
    from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor

    input_ids, embeds, mask = uv.prepare_gpt_inputs(conds_latent.unsqueeze(0), text_ids.unsqueeze(0))
    model = uv.inference_model
    model.store_mel_emb(embeds)
    proc = RepetitionPenaltyLogitsProcessor(theta)
    past = None
    ids, margins, logits_all = [], [], []
    for k in range(n_steps):
        inp = model.prepare_inputs_for_generation(input_ids, past_key_values=past, attention_mask=mask, use_cache=True)
        out = model(**inp, return_dict=True)
        logits = out.logits[:, -1, :].float()
        past = out.past_key_values
        scores = proc(input_ids, logits.clone())
        top2 = torch.topk(scores[0], 2).values
        margins.append(float(top2[0] - top2[1]))
        logits_all.append(logits[0].clone())
        tok = int(scores[0].argmax())
        if forced is not None:
            tok = int(forced[k])
        ids.append(tok)
        input_ids = torch.cat([input_ids, torch.tensor([[tok]])], dim=1)
        mask = torch.cat([mask, torch.ones(1, 1, dtype=mask.dtype)], dim=1)
    return ids, margins, torch.stack(logits_all), embeds[0], mask[0, : embeds.shape[1] + 1]


def gen_gpt():
    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    W = WR.make_gpt_weights(cfg, seed=31, head_scale=50.0)
    uv = build_ref_gpt(cfg, W)
    g = torch.Generator().manual_seed(32)
    D = cfg["model_dim"]
    cond32 = torch.randn(32, D, generator=g) * 0.5
    emo_vec = torch.randn(D, generator=g) * 0.5
    # inference_speech:693-696
    conds_latent = torch.cat((cond32 + emo_vec.unsqueeze(0), uv.speed_emb.weight[1:2], uv.speed_emb.weight[0:1]), 0)
    out = dict(seed=31, model_dim=D, layers=2, heads=2, cond32=cond32, emo_vec=emo_vec, conds_latent=conds_latent)

    for tag, text in (("plain", torch.randint(2, 200, (12,), generator=g)),
                      ("padded", torch.cat((torch.tensor([0, 0, 1]), torch.randint(2, 200, (9,), generator=g))))):
        text = text.to(torch.int32)
        n = 40
        ids, margins, logits, embeds, mask = ref_greedy(uv, conds_latent, text, n)
        out[f"text_{tag}"] = text
        out[f"embeds_{tag}"] = embeds
        out[f"mask_{tag}"] = mask
        out[f"ids_{tag}"] = np.array(ids)
        out[f"margins_{tag}"] = np.array(margins)
        out[f"logits_{tag}"] = logits[[0, 1, 2, n - 1]]
        print(tag, "ids", ids[:12], "min margin", min(margins))

    # latent pass (G9): UnifiedVoice.forward with given latent + emo_vec (model_v2.py:554-596)
    text = out["text_plain"].long().unsqueeze(0)
    codes = torch.tensor(out["ids_plain"][:25]).unsqueeze(0)
    lat = uv(cond32.unsqueeze(0), text.clone(), torch.tensor([text.shape[-1]]), codes.clone(), torch.tensor([codes.shape[-1]]),
             None, emo_vec=emo_vec.unsqueeze(0), use_speed=torch.zeros(1).long())
    out["latent_codes"] = codes[0]
    out["latent"] = lat[0]
    save("gpt_tiny.npz", **out)


def gen_gpt_block_prod():
    """One production-width decoder layer + head: weights regenerated from the seed on both sides."""
    cfg = dict(WR.GPT_CFG)
    cfg.update(layers=1, max_text_tokens=40, max_mel_tokens=80, number_text_tokens=200)
    W = WR.make_gpt_weights(cfg, seed=41, head_scale=50.0)
    uv = build_ref_gpt(cfg, W)
    g = torch.Generator().manual_seed(42)
    D = cfg["model_dim"]
    conds_latent = torch.randn(34, D, generator=g) * 0.5
    text = torch.randint(2, 200, (6,), generator=g).to(torch.int32)
    ids, margins, logits, embeds, mask = ref_greedy(uv, conds_latent, text, 6)
    save("gpt_prod_layer.npz", seed=41, conds_latent=conds_latent, text=text, ids=np.array(ids), margins=np.array(margins),
         logits_first=logits[0], logits_last=logits[-1])


def gen_beam():
    out = {}
    for tag, bias, seed, lp in (("noeos", 14.0, 61, 0.0), ("mid", 31.0, 64, 0.0), ("mid2", 34.0, 65, 0.0), ("eos", 38.0, 62, 0.0), ("eos2", 42.0, 63, 0.0),
                                ("lp1", 36.0, 66, 1.0), ("lpneg", 34.0, 67, -0.7), ("lp2noeos", 14.0, 68, 2.0)):
        r = _gen_beam_case(bias, seed, lp)
        print(tag, "steps", r["picks"].shape[0], "done", int(r["done"]), "seq", r["sequence"].tolist())
        out.update({f"{tag}_{k}": v for k, v in r.items()})
    save("gpt_beam.npz", seed=31, **out)


def gen_beam_search():
    """Beam search WITHOUT sampling (`_beam_search` with do_sample=False, transformers_generation_utils.py:3511-3524: the candidates are
    the top 2 * num_beams of the joint scores; only the processors run, the warpers are sampling-only, :1020)."""
    out = {}
    for tag, bias, seed, lp, nb in (("noeos", 4.0, 81, 0.0, 3), ("mid", 30.0, 82, 0.0, 3), ("mid2", 33.0, 87, 0.0, 3), ("eos", 38.0, 83, 0.0, 3), ("lp1", 33.0, 84, 1.0, 3),
                                    ("lp2", 30.0, 88, 2.0, 3), ("nb2", 32.0, 85, 0.0, 2), ("nb4", 32.0, 86, 0.0, 4)):
        r = _gen_beam_case(bias, seed, lp, do_sample=False, nb=nb)
        print(tag, "steps", r["picks"].shape[0], "done", int(r["done"]), "seq", r["sequence"].tolist())
        out.update({f"{tag}_{k}": v for k, v in r.items()})
    save("gpt_beam_search.npz", seed=31, **out)


def _gen_beam_case(stop_bias, rng_seed, length_penalty=0.0, do_sample=True, nb=3):
    """3-beam beam-sample through the reference's own BeamSearchScorer (vendored text,
    transformers_beam_search.py:123-417) + HF processors + the reference model forward; draws come from a
    seeded torch.multinomial and are stored so any implementation can replay them."""
    from indextts.gpt.transformers_beam_search import BeamSearchScorer
    from transformers.generation.logits_process import (
        RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)

    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    W = WR.make_gpt_weights(cfg, seed=31, head_scale=50.0)
    # make EOS reachable so hypotheses are produced: bias the stop token
    W["mel_head.bias"] = W["mel_head.bias"].clone()
    W["mel_head.bias"][8193] += stop_bias
    uv = build_ref_gpt(cfg, W)
    g = torch.Generator().manual_seed(rng_seed)
    D = cfg["model_dim"]
    conds_latent = torch.randn(34, D, generator=g) * 0.5
    text = torch.randint(2, 200, (10,), generator=g).to(torch.int32)
    V, max_new = 8194, 24
    input_ids, embeds, mask = uv.prepare_gpt_inputs(conds_latent.unsqueeze(0), text.unsqueeze(0))
    P = input_ids.shape[1]
    model = uv.inference_model
    model.store_mel_emb(embeds)
    procs = [RepetitionPenaltyLogitsProcessor(10.0), TemperatureLogitsWarper(0.8), TopKLogitsWarper(30, min_tokens_to_keep=2),
             TopPLogitsWarper(0.8, min_tokens_to_keep=2)]
    if not do_sample:
        procs = procs[:1]
    scorer = BeamSearchScorer(batch_size=1, num_beams=nb, device="cpu", length_penalty=length_penalty, do_early_stopping=False,
                              num_beam_hyps_to_keep=1, max_length=P + max_new)
    input_ids = input_ids.repeat_interleave(nb, dim=0)
    mask = mask.repeat_interleave(nb, dim=0)
    beam_scores = torch.zeros(nb)
    beam_scores[1:] = -1e9
    past = None
    all_picks, all_ns, all_nt, all_ni = [], [], [], []
    next_tokens = next_indices = None
    for step in range(max_new):
        inp = model.prepare_inputs_for_generation(input_ids, past_key_values=past, attention_mask=mask, use_cache=True)
        out = model(**inp, return_dict=True)
        past = out.past_key_values
        scores = torch.log_softmax(out.logits[:, -1, :].float(), dim=-1)
        for pr in procs:
            scores = pr(input_ids, scores)
        scores = scores + beam_scores[:, None]
        flat = scores.view(1, nb * V)
        if do_sample:
            picks = torch.multinomial(torch.softmax(flat, -1), 2 * nb, generator=g)
            sc = torch.gather(flat, -1, picks)
            sc, order = torch.sort(sc, descending=True, dim=1)
            picks = torch.gather(picks, -1, order)
        else:
            sc, picks = torch.topk(flat, 2 * nb, dim=1, largest=True, sorted=True)
        next_indices = torch.div(picks, V, rounding_mode="floor")
        next_tokens = picks % V
        bo = scorer.process(input_ids, sc, next_tokens, next_indices, pad_token_id=8193, eos_token_id=8193, decoder_prompt_len=P)
        beam_scores = bo["next_beam_scores"]
        bt, bi = bo["next_beam_tokens"], bo["next_beam_indices"]
        all_picks.append(picks[0].clone())
        all_ns.append(beam_scores.clone())
        all_nt.append(bt.clone())
        all_ni.append(bi.clone())
        input_ids = torch.cat([input_ids[bi, :], bt.unsqueeze(-1)], dim=-1)
        mask = torch.cat([mask, torch.ones(nb, 1, dtype=mask.dtype)], dim=1)
        if hasattr(past, "reorder_cache"):
            past.reorder_cache(bi)
        else:
            past = model._reorder_cache(past, bi)
        if scorer.is_done:
            break
    fin = scorer.finalize(input_ids, beam_scores, next_tokens, next_indices, pad_token_id=8193, eos_token_id=8193,
                          max_length=P + max_new, decoder_prompt_len=P)
    seq = fin["sequences"][0, P:]
    return dict(stop_bias=stop_bias, length_penalty=length_penalty, conds_latent=conds_latent, text=text, max_new=max_new, picks=torch.stack(all_picks),
                next_scores=torch.stack(all_ns), next_tokens=torch.stack(all_nt), next_indices=torch.stack(all_ni),
                sequence=seq, sequence_score=fin["sequence_scores"], done=int(bool(scorer.is_done)))


# ----------------------------------------------------------------------------- N1 (s2mel glue)
class _AD(dict):
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def _load_folded_into_reference(module, W, prefix):
    """Put our (weight-norm-folded) tensors into a reference module that still carries weight_g / weight_v."""
    sd = module.state_dict()
    new, used = {}, set()
    for k, v in sd.items():
        if k.endswith("weight_v") and prefix + k[: -len("weight_v")] + "weight" in W:
            w = W[prefix + k[: -len("weight_v")] + "weight"]
            new[k] = w
            used.add(prefix + k[: -len("weight_v")] + "weight")
        elif k.endswith("weight_g") and prefix + k[: -len("weight_g")] + "weight" in W:
            w = W[prefix + k[: -len("weight_g")] + "weight"]
            new[k] = w.reshape(w.shape[0], -1).norm(dim=1).reshape(v.shape)
        elif prefix + k in W:
            assert tuple(W[prefix + k].shape) == tuple(v.shape), (k, W[prefix + k].shape, v.shape)
            new[k] = W[prefix + k]
            used.add(prefix + k)
        else:
            new[k] = v
    module.load_state_dict(new, strict=True)
    return used


def build_ref_s2mel(cfg, W, max_T=64):
    """The reference's MyModel (gpt_layer, length_regulator, CFM/DiT/WaveNet) + FactorizedVectorQuantize with our tensors."""
    from indextts.s2mel.modules.commons import MyModel
    from indextts.utils.maskgct.models.codec.amphion_codec.quantize.factorized_vector_quantize import FactorizedVectorQuantize

    args = _AD(
        reg_loss_type="l1", dit_type="DiT", style_encoder=_AD(dim=cfg["style_dim"]),
        length_regulator=_AD(channels=cfg["lr_channels"], is_discrete=False, in_channels=cfg["lr_in_channels"], content_codebook_size=2048,
                             sampling_ratios=[1] * cfg["lr_n_blocks"], vector_quantize=False, n_codebooks=1, quantizer_dropout=0.0,
                             f0_condition=False, n_f0_bins=512),
        DiT=_AD(hidden_dim=cfg["hidden_dim"], num_heads=cfg["num_heads"], depth=cfg["depth"], class_dropout_prob=0.1, block_size=8192,
                in_channels=80, style_condition=True, final_layer_type="wavenet", content_dim=cfg["content_dim"],
                content_codebook_size=1024, content_type="discrete", is_causal=False, long_skip_connection=True,
                zero_prompt_speech_token=False, time_as_token=False, style_as_token=False, uvit_skip_connection=True),
        wavenet=_AD(hidden_dim=cfg["wavenet_hidden"], num_layers=cfg["wavenet_layers"], kernel_size=cfg["wavenet_kernel"],
                    dilation_rate=cfg["wavenet_dilation_rate"], p_dropout=0.2, style_condition=True),
    )
    m = MyModel(args, use_gpt_latent=True).eval()
    m.models["cfm"].estimator.setup_caches(max_batch_size=2, max_seq_length=max(64, max_T))
    used = set()
    for key in ("cfm", "length_regulator", "gpt_layer"):
        used |= _load_folded_into_reference(m.models[key], W, key + ".")
    q = FactorizedVectorQuantize(input_dim=cfg["semantic_dim"], codebook_size=cfg["codebook_size"], codebook_dim=cfg["codebook_dim"]).eval()
    used |= _load_folded_into_reference(q, W, "quantizer.")
    missing = set(W) - used
    assert not missing, sorted(missing)[:5]
    return m, q


def gen_s2mel(name="s2mel_tiny.npz", seed=71, n=5, Tp=7, **cfg_kw):
    """`s2mel_tiny.npz`: hidden 64 / 2 heads (head_dim 32).  `s2mel_hd64.npz`: hidden 128 / 2 heads -> head_dim 64, the
    production head size, so the GPU test of this fixture runs through `attn_full_f32_kernel` (s2mel.py `_attention`), and
    long enough (T = 70 + 154) to cross the kernel's 32-query / 64-key tile edges."""
    import voice_tts_amd.s2mel as S2

    cfg = S2.tiny_s2mel_cfg(gpt_dim=1280, semantic_dim=1024, lr_in_channels=1024, codebook_size=8194, **cfg_kw)
    W = S2.make_s2mel_weights(cfg, seed=seed)
    m, q = build_ref_s2mel(cfg, W, Tp + int(n * 1.72))

    g = torch.Generator().manual_seed(seed + 1)
    latent = torch.randn(1, n, 1280, generator=g) * 0.5
    codes = torch.randint(0, 8192, (1, n), generator=g)
    code_lens = torch.tensor([n])
    prompt_condition = torch.randn(1, Tp, cfg["content_dim"], generator=g)
    ref_mel = torch.randn(1, 80, Tp, generator=g) * 2 - 4
    style = torch.randn(1, cfg["style_dim"], generator=g)
    # infer_v2.py:713-731 with the reference's own modules
    lat = m.models["gpt_layer"](latent)
    S_infer = q.vq2emb(codes).transpose(1, 2) + lat
    target_lengths = (code_lens * 1.72).long()
    cond = m.models["length_regulator"](S_infer, ylens=target_lengths, n_quantizers=3, f0=None)[0]
    cat_condition = torch.cat([prompt_condition, cond], dim=1)
    T = cat_condition.shape[1]
    noise = torch.randn(1, 80, T, generator=g)
    cfm = m.models["cfm"]
    t_span = torch.linspace(0, 1, 4)
    with torch.inference_mode():
        mel = cfm.solve_euler(noise.clone(), torch.LongTensor([T]), ref_mel, cat_condition.clone(), style, None, t_span, inference_cfg_rate=0.7)
        one = cfm.estimator(noise, torch.zeros_like(noise), torch.LongTensor([T]), torch.tensor([0.3]), style, cat_condition)
    save(name, seed=seed, latent=latent, codes=codes, prompt_condition=prompt_condition, ref_mel=ref_mel, style=style,
         noise=noise, gpt_layer_out=lat, vq_emb=q.vq2emb(codes).transpose(1, 2), cond=cond, dit_one=one, mel=mel[:, :, Tp:], n_steps=3)


# ----------------------------------------------------------------------------- chain (SURVEY 8(d) config 1 on twins)
def gen_chain():
    """The reference's own segment chain on small twins (infer_v2.py:641-744): UnifiedVoice greedy decode (harness loop,
    SURVEY F5) -> UnifiedVoice.forward latent -> MyModel gpt_layer + vq2emb + length_regulator + CFM (noise injected,
    SURVEY F9) -> BigVGAN -> clamp(32767 x) -> int16.  Every stage output is stored, so a chained implementation is held to
    the END of the chain (waveform / PCM) with its own intermediate values, not stage by stage on fresh inputs."""
    import voice_tts_amd.s2mel as S2

    # MyModel.gpt_layer is fixed 1280 -> 256 -> 128 -> 1024 (commons.py:420-424): the GPT twin keeps the production width
    gcfg = WR.tiny_gpt_cfg(model_dim=1280, layers=2, heads=20)
    Wg = WR.make_gpt_weights(gcfg, seed=91, head_scale=50.0)
    uv = build_ref_gpt(gcfg, Wg)
    scfg = S2.tiny_s2mel_cfg(gpt_dim=1280, semantic_dim=1024, lr_in_channels=1024, codebook_size=8194, hidden_dim=128, num_heads=2,
                             wavenet_hidden=128, depth=3)  # head_dim 64: the HIP attention kernel is on the chain
    Ws = S2.make_s2mel_weights(scfg, seed=92)
    bcfg = WR.tiny_bigvgan_cfg(64)
    Wb = WR.make_bigvgan_weights(bcfg, seed=93)
    g = torch.Generator().manual_seed(94)
    D = gcfg["model_dim"]
    cond32 = torch.randn(32, D, generator=g) * 0.5
    emo_vec = torch.randn(D, generator=g) * 0.5
    conds_latent = torch.cat((cond32 + emo_vec.unsqueeze(0), uv.speed_emb.weight[1:2], uv.speed_emb.weight[0:1]), 0)
    text = torch.randint(2, 200, (14,), generator=g).to(torch.int32)
    n, Tp = 48, 30
    ids, margins, _, _, _ = ref_greedy(uv, conds_latent, text, n)
    assert 8193 not in ids
    codes = torch.tensor(ids).unsqueeze(0)
    t = text.long().unsqueeze(0)
    latent = uv(cond32.unsqueeze(0), t.clone(), torch.tensor([t.shape[-1]]), codes.clone(), torch.tensor([n]), None,
                emo_vec=emo_vec.unsqueeze(0), use_speed=torch.zeros(1).long())
    T = Tp + int(n * 1.72)
    m, q = build_ref_s2mel(scfg, Ws, T)
    prompt_condition = torch.randn(1, Tp, scfg["content_dim"], generator=g)
    ref_mel = (torch.randn(1, 80, Tp, generator=g) * 2 - 4).clamp(-11.5, 2)
    style = torch.randn(1, scfg["style_dim"], generator=g)
    noise = torch.randn(1, 80, T, generator=g)
    # infer_v2.py:713-731
    lat = m.models["gpt_layer"](latent)
    S_infer = q.vq2emb(codes).transpose(1, 2) + lat
    cond = m.models["length_regulator"](S_infer, ylens=(torch.tensor([n]) * 1.72).long(), n_quantizers=3, f0=None)[0]
    cat_condition = torch.cat([prompt_condition, cond], dim=1)
    n_steps = 4
    with torch.inference_mode():
        mel = m.models["cfm"].solve_euler(noise.clone(), torch.LongTensor([T]), ref_mel, cat_condition.clone(), style, None,
                                          torch.linspace(0, 1, n_steps + 1), inference_cfg_rate=0.7)[:, :, Tp:]
    voc = build_ref_bigvgan(bcfg, Wb)
    wav = voc(mel.float()).squeeze().unsqueeze(0)  # infer_v2.py:735-736
    wav = torch.clamp(32767 * wav, -32767.0, 32767.0)  # :740
    pcm = wav.type(torch.int16)  # :781
    print("chain: ids", ids[:10], "min margin", min(margins), "mel range", float(mel.min()), float(mel.max()), "wav max", float(wav.abs().max()))
    save("chain_tiny.npz", seeds=np.array([91, 92, 93]), cond32=cond32, emo_vec=emo_vec, conds_latent=conds_latent, text=text, ids=np.array(ids),
         margins=np.array(margins), latent=latent[0], prompt_condition=prompt_condition, ref_mel=ref_mel, style=style, noise=noise, n_steps=n_steps,
         mel=mel, wav=wav, pcm=pcm)


# ----------------------------------------------------------------------------- prompt-side glue (infer_v2.py:508-580)
def gen_prompt():
    """Reference classes of the once-per-prompt stages with our seeded tensors loaded: CAMPPlus (campplus/DTDNN.py), RepCodec.quantize
    (repcodec_model.py:176-196) and s2mel/modules/audio.py `mel_spectrogram`.  librosa is absent, so the mel BASIS handed to the
    reference function is ours (`slaney_mel_basis`): the fixture pins the padding / STFT / magnitude / log part, not the basis."""
    import voice_tts_amd.prompt as PR
    from indextts.s2mel.modules import audio as RA
    from indextts.s2mel.modules.campplus.DTDNN import CAMPPlus
    from indextts.utils.maskgct.models.codec.kmeans.repcodec_model import RepCodec

    g = torch.Generator().manual_seed(101)
    out = {}
    Wc = PR.make_camplus_weights(seed=102)
    cam = CAMPPlus(feat_dim=80, embedding_size=192).eval()
    sd = cam.state_dict()
    assert set(Wc) == {k for k in sd if not k.endswith("num_batches_tracked")}
    cam.load_state_dict({k: Wc.get(k, v) for k, v in sd.items()}, strict=True)
    for T in (215, 57):
        feat = torch.randn(1, T, 80, generator=g) * 2.0
        feat = feat - feat.mean(dim=1, keepdim=True)
        out[f"cam_feat_{T}"] = feat
        out[f"cam_style_{T}"] = cam(feat)
    ccfg = dict(codebook_size=50, hidden_size=48, codebook_dim=8, vocos_dim=32, vocos_intermediate_dim=64, vocos_num_layers=3)
    Wq = PR.make_codec_weights(ccfg, seed=103)
    codec = RepCodec(**ccfg).eval()
    used = _load_folded_into_reference(codec, Wq, "")
    assert not (set(Wq) - used), sorted(set(Wq) - used)[:5]
    x = torch.randn(2, 33, 48, generator=g)
    codes, S_ref = codec.quantize(x)
    out.update(codec_x=x, codec_codes=codes, codec_S=S_ref, codec_cfg=np.array(list(ccfg.values())))
    RA.librosa_mel_fn = lambda sr, n_fft, n_mels, fmin, fmax: PR.slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax)
    y = torch.randn(1, 22050, generator=g) * 0.1
    out.update(mel_y=y, mel_out=RA.mel_spectrogram(y, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0, fmax=None, center=False))
    save("prompt_tiny.npz", seeds=np.array([102, 103]), **out)


# ----------------------------------------------------------------------------- N2 (conditioning encoders)
def gen_conditioning():
    """Reference ConformerEncoder + PerceiverResampler composed exactly as UnifiedVoice.get_conditioning / get_emovec /
    merge_emovec compose them (model_v2.py:349-377,513-549,736-747), on a small twin, with our seeded weights loaded."""
    import torch.nn as nn
    import voice_tts_amd.conditioning as CD
    from indextts.gpt.conformer_encoder import ConformerEncoder
    from indextts.gpt.perceiver import PerceiverResampler

    cfg = CD.tiny_cond_cfg()
    W = CD.make_cond_weights(cfg, seed=81)
    cm, em, D = cfg["condition_module"], cfg["emo_condition_module"], cfg["model_dim"]
    mk_enc = lambda m: ConformerEncoder(input_size=cfg["input_size"], output_size=m["output_size"], linear_units=m["linear_units"],
                                        attention_heads=m["attention_heads"], num_blocks=m["num_blocks"], input_layer="conv2d2").eval()
    enc, emo_enc = mk_enc(cm), mk_enc(em)
    perc = PerceiverResampler(D, dim_context=cm["output_size"], ff_mult=cm["perceiver_mult"], heads=cm["attention_heads"],
                              num_latents=cfg["cond_num"], dim_head=cfg["perceiver_dim_head"]).eval()
    emo_perc = PerceiverResampler(cfg["emo_dim"], dim_context=em["output_size"], ff_mult=em["perceiver_mult"], heads=em["attention_heads"],
                                  num_latents=1, dim_head=cfg["perceiver_dim_head"]).eval()
    emovec_layer, emo_layer = nn.Linear(cfg["emo_dim"], D), nn.Linear(D, D)
    used = set()
    for mod, pre in ((enc, "conditioning_encoder."), (perc, "perceiver_encoder."), (emo_enc, "emo_conditioning_encoder."),
                     (emo_perc, "emo_perceiver_encoder."), (emovec_layer, "emovec_layer."), (emo_layer, "emo_layer.")):
        used |= _load_folded_into_reference(mod, W, pre)
    assert not (set(W) - used), sorted(set(W) - used)[:5]
    pad_c, pad_e = nn.ConstantPad1d((cfg["cond_num"], 0), True), nn.ConstantPad1d((1, 0), True)

    def get_conditioning(x, lens):  # x [B, idim, T]
        h, mask = enc(x.transpose(1, 2), lens)
        return perc(h, pad_c(mask.squeeze(1)))

    def get_emovec(x, lens):  # x [B, T, idim]
        h, mask = emo_enc(x, lens)
        v = emo_perc(h, pad_e(mask.squeeze(1))).squeeze(1)
        return emo_layer(emovec_layer(v))

    g = torch.Generator().manual_seed(82)
    T = 23
    spk = torch.randn(2, T, cfg["input_size"], generator=g)
    emo = torch.randn(2, T - 4, cfg["input_size"], generator=g)
    full, ragged = torch.tensor([T, T]), torch.tensor([T, T - 6])
    with torch.inference_mode():
        enc_full, mask_full = enc(spk, full)
        enc_rag, mask_rag = enc(spk, ragged)
        cond_full = get_conditioning(spk.transpose(1, 2), full)
        cond_rag = get_conditioning(spk.transpose(1, 2), ragged)
        ev_spk, ev_emo = get_emovec(spk, full), get_emovec(emo, torch.tensor([T - 4, T - 4]))
        merged = ev_spk + 0.7 * (ev_emo - ev_spk)
    save("conditioning_tiny.npz", seed=81, spk=spk, emo=emo, lens_ragged=ragged, enc_full=enc_full, enc_ragged=enc_rag, mask_ragged=mask_rag,
         cond_full=cond_full, cond_ragged=cond_rag, emovec_spk=ev_spk, emovec_emo=ev_emo, merged_alpha07=merged)


# ----------------------------------------------------------------------------- G8
def gen_sampler():
    from transformers.generation.logits_process import (
        RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)

    g = torch.Generator().manual_seed(51)
    V = 8194
    out = {}
    for case, min_keep in (("sample", 1), ("beam", 2)):
        logits = torch.randn(1, V, generator=g) * 3.0
        hist = torch.cat((torch.ones(1, 30, dtype=torch.long), torch.tensor([[8192]]), torch.randint(0, 8192, (1, 50), generator=g)), dim=1)
        s0 = logits if case == "sample" else torch.log_softmax(logits, dim=-1)
        s1 = RepetitionPenaltyLogitsProcessor(10.0)(hist, s0.clone())
        s2 = TemperatureLogitsWarper(0.8)(hist, s1.clone())
        s3 = TopKLogitsWarper(30, min_tokens_to_keep=min_keep)(hist, s2.clone())
        s4 = TopPLogitsWarper(0.8, min_tokens_to_keep=min_keep)(hist, s3.clone())
        out[f"{case}_in"] = s0[0]
        out[f"{case}_hist"] = hist[0]
        out[f"{case}_pen"] = s1[0]
        out[f"{case}_final"] = s4[0]
        out[f"{case}_probs"] = torch.softmax(s4[0], -1)
        # the same chain with the reference's custom typical-sampling processor after the penalty (model_v2.py:717-722)
        from indextts.utils.typical_sampling import TypicalLogitsWarper
        for mass in (0.9, 0.3):
            t1 = TypicalLogitsWarper(mass=mass, min_tokens_to_keep=min_keep)(hist, s1.clone())
            t4 = TopPLogitsWarper(0.8, min_tokens_to_keep=min_keep)(hist, TopKLogitsWarper(30, min_tokens_to_keep=min_keep)(hist, TemperatureLogitsWarper(0.8)(hist, t1.clone())))
            tag = f"{case}_typ{int(mass * 100)}"
            out[tag + "_kept"] = torch.isfinite(t1[0])
            out[tag + "_probs"] = torch.softmax(t4[0], -1)
        # G8 corners generate() can reach through infer(top_k=..., top_p=...): top_k beyond 128, top_k = 0 (HF then builds no
        # TopK warper, generation_utils.py:1035-1038), top_p = 1.0 (no TopP warper), a top_p so small only the best survives
        for wk, wp, wt in ((0, 0.8, 0.8), (500, 0.95, 0.9), (129, 0.6, 0.8), (0, 1.0, 1.3), (0, 0.02, 0.8), (4000, 0.999, 1.0)):
            w = TemperatureLogitsWarper(wt)(hist, s1.clone()) if wt != 1.0 else s1.clone()
            if wk > 0:
                w = TopKLogitsWarper(wk, min_tokens_to_keep=min_keep)(hist, w)
            if wp < 1.0:
                w = TopPLogitsWarper(wp, min_tokens_to_keep=min_keep)(hist, w)
            out[f"{case}_wide_k{wk}_p{int(wp * 1000)}_t{int(wt * 10)}"] = torch.softmax(w[0], -1)
    save("sampler_kat.npz", **out)


FRONT_TEXTS = [
    # inputs from the reference's own case list (front.py:440-489) that need no number verbaliser, plus a few of ours
    "晕XUAN4是一种GAN3觉", "我爱你！", "I love you!", "“我爱你”的英语是“I love you”", "受不liao3你了",
    "“衣裳”不读衣chang2，而是读衣shang5", "最zhong4要的是：不要chong2蹈覆辙", "不zuo1死就不会死", "这酒...里...有毒...",
    "只有,,,才是最好的", "只有，，，才是最好的", "用beta1测试", "have you ever been to beta2?", "where's the money?", "who's there?",
    "which's the best?", "how's it going?", "今天是个好日子 it's a good day",
    "约瑟夫·高登-莱维特（Joseph Gordon-Levitt is an American actor）",
    "蒂莫西·唐纳德·库克（英文名：Timothy Donald Cook），通称蒂姆·库克（Tim Cook），美国商业经理、工业工程师和工业开发商，现任苹果公司首席执行官。",
    "《盗梦空间》是由美国华纳兄弟影片公司出品的电影，由克里斯托弗·诺兰执导并编剧，莱昂纳多·迪卡普里奥、玛丽昂·歌迪亚等联袂主演。",
    "ju4 que4 xün1 lüe4 nv3 是拼音", "someone@example.com", "such as XTTS, CosyVoice, Fish-Speech, and F-TTS", "ta shuo: “ni hao3 ma5？”\n我说：好！  ",
    "price is $5 (five) ~ [six]; 「seven」—eight", "here's what she's saying: that's it", "",
]


def gen_front():
    """Text front-end: the reference's TextNormalizer (identity objects in place of the absent WeText verbalisers),
    its CJK helpers, TextTokenizer on a small sentencepiece model trained here, and its segment splitter on seeded
    random token streams (front.py:11-436, common.py:29-82)."""
    import json
    import random
    import warnings

    import sentencepiece as spm
    from indextts.utils.common import de_tokenized_by_CJK_char, tokenize_by_CJK_char
    from indextts.utils.front import TextNormalizer, TextTokenizer

    class Identity:
        def normalize(self, text):
            return text

    norm = TextNormalizer()
    norm.zh_normalizer, norm.en_normalizer = Identity(), Identity()
    out = {"normalize": [], "cjk": [], "pinyin": [], "tokenizer": [], "split": []}
    for t in FRONT_TEXTS:
        out["normalize"].append({"text": t, "use_chinese": norm.use_chinese(t), "normalized": norm.normalize(t)})
        spaced = tokenize_by_CJK_char(t)
        out["cjk"].append({"text": t, "spaced": spaced, "spaced_keep_case": tokenize_by_CJK_char(t, do_upper_case=False),
                           "joined": de_tokenized_by_CJK_char(spaced), "joined_lower": de_tokenized_by_CJK_char(spaced, do_lower_case=True)})
    for p in ["ju4", "que2", "xün1", "xuan4", "Juan3", "lü4", "zhong1", "QU5", "jue2", "xu1"]:
        out["pinyin"].append({"pinyin": p, "corrected": norm.correct_pinyin(p)})

    # a small BPE model of our own over the normalised, CJK-spaced, upper-cased case texts
    model_prefix = os.path.join(HERE, "tiny_bpe")
    corpus = os.path.join(HERE, "_tiny_bpe_corpus.txt")
    with open(corpus, "w", encoding="utf-8") as f:
        for rep in range(4):
            for t in FRONT_TEXTS:
                if t:
                    f.write(tokenize_by_CJK_char(norm.normalize(t)) + "\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=model_prefix, vocab_size=330, model_type="bpe", character_coverage=1.0,
                                   bos_id=0, eos_id=1, unk_id=2, pad_id=-1, user_defined_symbols=["XUAN4", "GAN3", "ZHONG4", "JV4"],
                                   normalization_rule_name="identity", minloglevel=2)
    os.remove(corpus)
    os.remove(model_prefix + ".vocab")
    tok = TextTokenizer(model_prefix + ".model", norm)
    out["tokenizer_meta"] = {"vocab_size": tok.vocab_size, "unk_token_id": tok.unk_token_id, "special_tokens_map": tok.special_tokens_map,
                             "punct_ids": tok.convert_tokens_to_ids(list(TextTokenizer.punctuation_marks_tokens))}
    for t in FRONT_TEXTS + ["好", " a ", "未登录字符：龘"]:
        ids = tok.encode(t)
        toks = tok.tokenize(t)
        out["tokenizer"].append({"text": t, "ids": ids, "tokens": toks, "decoded": tok.decode(ids) if ids else "",
                                 "decoded_lower": tok.decode(ids, do_lower_case=True) if ids else "",
                                 "segments_20": tok.split_segments(toks, 20), "segments_120": tok.split_segments(toks)})
    out["tokenizer_batch"] = tok.batch_encode([t for t in FRONT_TEXTS if t][:6])

    rng = random.Random(20251128)
    words = ["▁A", "B", "▁CD", "你", "好", "▁E", "F", "世", "界", "▁XUAN4"]
    marks = [".", "!", "?", "▁.", "▁?", "▁...", ",", "▁,", "-", "'", "▁'", "…"]
    for case in range(400):
        n = rng.choice([0, 1, 2, 3, 7, 15, 40, 90, 200])
        p_mark = rng.choice([0.02, 0.08, 0.2, 0.4])
        toks = [rng.choice(marks) if rng.random() < p_mark else rng.choice(words) for _ in range(n)]
        limit = rng.choice([4, 8, 20, 50, 120])
        quick = rng.choice([0, 0, 10, 60])
        rec = {"tokens": toks, "limit": limit, "quick": quick}
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            try:
                rec["segments"] = TextTokenizer.split_segments_by_token(toks, list(TextTokenizer.punctuation_marks_tokens), limit, quick)
            except (AssertionError, RecursionError) as e:
                rec["error"] = type(e).__name__
            rec["warned"] = any(issubclass(w.category, RuntimeWarning) for w in caught)
        out["split"].append(rec)
    with open(os.path.join(HERE, "front.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, separators=(",", ":"))
    print("front.json:", {k: (len(v) if isinstance(v, list) else "-") for k, v in out.items()},
          "split errors", sum("error" in r for r in out["split"]), "warned", sum(r["warned"] for r in out["split"]))


def gen_emotion():
    """The /tts emotion-label vocabulary and known answers of the reference's emotion.py (imports as-is)."""
    import json

    import emotion as RE  # /root/reference/emotion.py

    probes = ["happy", " Joyful ", "ANXIOUS", "紧张", "normal", "glad", "rage", "no-such-label", "", "惊喜", "stunned"]
    dicts = [{"高兴": 0.7, "平静": 0.3}, {"happy": 0.7, "joyful": 0.5}, {"开心": 0.8, "生气": 0.2}, {"anxious": 0.4, "panic": 0.9, "??": 0.6},
             {"downcast": 1.0, "低沉": 0.2, "calmness": 0.1}, {}]
    out = {"order": RE.STANDARD_EMOTION_ORDER, "mapping": RE.EMOTION_MAPPING,
           "label_cases": [{"label": p, "standard": RE.normalize_emotion_label(p)} for p in probes],
           "string_cases": [{"label": p, "alpha": a, "vector": RE.create_emotion_vector(p, a)} for p in probes for a in (1.0, 0.35)],
           "dict_cases": [{"input": d, "vector": RE.create_emotion_vector(d)} for d in dicts]}
    with open(os.path.join(HERE, "emotion_labels.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("emotion_labels.json:", len(out["mapping"]), "labels")


if __name__ == "__main__":
    which = sys.argv[1:] or ["emotion", "aa", "bigvgan", "gpt", "prod", "sampler", "beam", "beamsearch", "s2mel", "chain", "prompt", "cond", "front"]
    if "emotion" in which:
        gen_emotion()
    if "aa" in which:
        gen_aa_snake()
    if "bigvgan" in which:
        gen_bigvgan()
    if "gpt" in which:
        gen_gpt()
    if "prod" in which:
        gen_gpt_block_prod()
    if "sampler" in which:
        gen_sampler()
    if "beam" in which:
        gen_beam()
    if "beamsearch" in which:
        gen_beam_search()
    if "s2mel" in which:
        gen_s2mel()
        gen_s2mel("s2mel_hd64.npz", seed=73, n=90, Tp=70, hidden_dim=128, num_heads=2, wavenet_hidden=128, depth=3)
    if "chain" in which:
        gen_chain()
    if "prompt" in which:
        gen_prompt()
    if "cond" in which:
        gen_conditioning()
    if "front" in which:
        gen_front()
