"""Pin oracle/ (our CPU restatement) against fixtures produced by the reference's own modules.

tests/golden/*.npz come from tests/golden/make_golden.py (reference classes, seeded weights).
"""
import numpy as np
import pytest
import torch

from oracle import gpt as OG
from oracle import vocoder as OV
import voice_tts_amd.weights as WR


def test_filter_taps(golden):
    g = golden("aa_snake.npz")
    f = OV.kaiser_sinc_filter12()
    assert np.array_equal(f, g["up_filter"])
    assert np.array_equal(f, g["down_filter"])
    assert abs(float(f.sum()) - 1.0) < 1e-6


@pytest.mark.parametrize("tag", ["c3_t1", "c3_t2", "c5_t5", "c4_t11", "c24_t64", "c6_t300", "c2_t4097"])
def test_aa_snake_vs_reference(golden, tag):
    g = golden("aa_snake.npz")
    y = OV.aa_snake(torch.from_numpy(g[f"x_{tag}"]), g[f"la_{tag}"], g[f"lb_{tag}"])
    ref = torch.from_numpy(g[f"y_{tag}"])
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())


def test_aa_snake_empty():
    y = OV.aa_snake(torch.zeros(1, 3, 0), np.zeros(3, np.float32), np.zeros(3, np.float32))
    assert y.shape == (1, 3, 0)


@pytest.mark.parametrize("F", [1, 7, 40])
def test_bigvgan_tiny_vs_reference(golden, F):
    g = golden("bigvgan_tiny.npz")
    cfg = WR.tiny_bigvgan_cfg(int(g["upsample_initial_channel"]))
    W = WR.make_bigvgan_weights(cfg, seed=int(g["seed"]))
    wav = OV.bigvgan_forward(torch.from_numpy(g[f"mel_f{F}"]), W, cfg)
    ref = torch.from_numpy(g[f"wav_f{F}"])
    assert wav.shape == ref.shape == (1, 1, 256 * F)
    assert ref.abs().max() > 0.05  # the fixture is not degenerate
    assert (ref.abs() >= 1.0).float().mean() < 0.2  # mostly unclamped
    assert (wav - ref).abs().max().item() <= 2e-5


def test_fold_weight_norm_matches_torch():
    torch.manual_seed(0)
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 5, 3))
    conv.weight_g.data *= 1.0 + torch.rand_like(conv.weight_g)
    convt = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(6, 4, 4, 2))
    sd = {"a." + k: v for k, v in conv.state_dict().items()}
    sd.update({"b." + k: v for k, v in convt.state_dict().items()})
    x = torch.randn(1, 6, 9)
    want_a, want_b = conv(x), convt(x)
    for fold in (WR.fold_weight_norm, OV.fold_weight_norm):
        f = fold(sd)
        assert set(f) == {"a.weight", "a.bias", "b.weight", "b.bias"}
        got_a = torch.nn.functional.conv1d(x, f["a.weight"], f["a.bias"])
        got_b = torch.nn.functional.conv_transpose1d(x, f["b.weight"], f["b.bias"], stride=2)
        assert torch.allclose(got_a, want_a, atol=1e-6) and torch.allclose(got_b, want_b, atol=1e-6)
    # parametrized spelling
    conv2 = torch.nn.utils.parametrizations.weight_norm(torch.nn.Conv1d(6, 5, 3))
    f2 = WR.fold_weight_norm({"c." + k: v for k, v in conv2.state_dict().items()})
    assert torch.allclose(torch.nn.functional.conv1d(x, f2["c.weight"], f2["c.bias"]), conv2(x), atol=1e-6)


def _tiny_oracle(g):
    cfg = WR.tiny_gpt_cfg(model_dim=int(g["model_dim"]), layers=int(g["layers"]), heads=int(g["heads"]))
    W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
    return OG.GptOracle(W, cfg["layers"], cfg["heads"]), cfg


@pytest.mark.parametrize("tag", ["plain", "padded"])
def test_gpt_prompt_and_greedy_vs_reference(golden, tag):
    g = golden("gpt_tiny.npz")
    orc, cfg = _tiny_oracle(g)
    cl = orc.conds_latent(g["cond32"], g["emo_vec"])
    assert torch.allclose(cl, torch.from_numpy(g["conds_latent"]), atol=0, rtol=0)
    fake, embeds, mask = orc.prepare_gpt_inputs(cl, g[f"text_{tag}"])
    assert torch.equal(mask, torch.from_numpy(g[f"mask_{tag}"]))
    assert torch.allclose(embeds, torch.from_numpy(g[f"embeds_{tag}"]), atol=1e-7)
    assert fake[-1] == 8192 and (fake[:-1] == 1).all()
    if tag == "padded":
        assert mask[:3].sum() == 0 and mask[3:].all()
    ref_ids = g[f"ids_{tag}"].tolist()
    n = len(ref_ids)
    ids, margins, logits = OG.generate_greedy(orc, embeds, mask, n, return_logits=True)
    assert ids == ref_ids
    assert np.allclose(margins, g[f"margins_{tag}"], atol=2e-3)
    ref_logits = torch.from_numpy(g[f"logits_{tag}"])
    got = logits[[0, 1, 2, n - 1]]
    assert (got - ref_logits).abs().max().item() <= 5e-4 * ref_logits.abs().max().item()
    # the one-pass teacher-forced form (what the full-size GPU tests check the device against) == the cached step loop,
    # and through it the reference's own forward: same logits at every step, same greedy choices (left padding included)
    one = OG.teacher_forced_logits(orc, embeds, mask, ref_ids)
    assert one.shape == (n + 1, logits.shape[1])
    assert (one[:n] - logits).abs().max().item() <= 2e-4 * logits.abs().max().item()
    assert (one[[0, 1, 2, n - 1]] - ref_logits).abs().max().item() <= 5e-4 * ref_logits.abs().max().item()
    picks, m2 = OG.greedy_choices(one, len(mask), ref_ids)
    assert picks == ref_ids and np.allclose(m2, g[f"margins_{tag}"], atol=2e-3)


def test_gpt_latent_pass_vs_reference(golden):
    g = golden("gpt_tiny.npz")
    orc, cfg = _tiny_oracle(g)
    lat = orc.latent_pass(torch.from_numpy(g["conds_latent"]), g["text_plain"], g["latent_codes"])
    ref = torch.from_numpy(g["latent"])
    assert lat.shape == ref.shape == (25, cfg["model_dim"])
    assert (lat - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_gpt_production_width_layer_vs_reference(golden):
    g = golden("gpt_prod_layer.npz")
    cfg = dict(WR.GPT_CFG)
    cfg.update(layers=1, max_text_tokens=40, max_mel_tokens=80, number_text_tokens=200)
    W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
    orc = OG.GptOracle(W, 1, cfg["heads"])
    fake, embeds, mask = orc.prepare_gpt_inputs(torch.from_numpy(g["conds_latent"]), g["text"])
    ids, margins, logits = OG.generate_greedy(orc, embeds, mask, 6, return_logits=True)
    assert ids == g["ids"].tolist()
    for got, ref in ((logits[0], g["logits_first"]), (logits[-1], g["logits_last"])):
        ref = torch.from_numpy(ref)
        assert (got - ref).abs().max().item() <= 5e-5 * ref.abs().max().item()


@pytest.mark.parametrize("case,min_keep", [("sample", 1), ("beam", 2)])
def test_sampler_corner_settings_vs_hf(golden, case, min_keep):
    """top_k = 0 (no TopK warper), top_k beyond 128, top_p = 1.0 (no TopP warper), a top_p that leaves min_tokens_to_keep: the
    probability vector the HF warpers produce for each setting (tests/golden/sampler_kat.npz, `*_wide_*`)."""
    g = golden("sampler_kat.npz")
    s_in = torch.from_numpy(g[f"{case}_in"])
    hist = g[f"{case}_hist"].tolist()
    keys = [k for k in g.files if k.startswith(f"{case}_wide_")]
    assert len(keys) == 6
    for key in keys:
        k, p, t = (int(x[1:]) for x in key.split("_")[2:])
        fin = OG.process_logits(s_in, hist, 10.0, t / 10.0, k, p / 1000.0, min_keep)
        ref = torch.from_numpy(g[key])
        got = torch.softmax(fin, -1)
        assert torch.equal(got > 0, ref > 0), key
        assert torch.allclose(got, ref, atol=1e-6), key


@pytest.mark.parametrize("case,min_keep", [("sample", 1), ("beam", 2)])
def test_sampler_processors_vs_hf(golden, case, min_keep):
    g = golden("sampler_kat.npz")
    s_in = torch.from_numpy(g[f"{case}_in"])
    hist = g[f"{case}_hist"].tolist()
    pen = OG.repetition_penalty(s_in, hist, 10.0)
    assert torch.equal(pen, torch.from_numpy(g[f"{case}_pen"]))
    fin = OG.process_logits(s_in, hist, 10.0, 0.8, 30, 0.8, min_keep)
    ref = torch.from_numpy(g[f"{case}_final"])
    assert torch.equal(torch.isinf(fin), torch.isinf(ref))
    keep = ~torch.isinf(ref)
    assert torch.allclose(fin[keep], ref[keep], atol=1e-6)
    assert torch.allclose(torch.softmax(fin, -1), torch.from_numpy(g[f"{case}_probs"]), atol=1e-6)
    assert int(keep.sum()) >= min_keep


@pytest.mark.parametrize("case,min_keep", [("sample", 1), ("beam", 2)])
@pytest.mark.parametrize("mass", [0.9, 0.3])
def test_typical_sampling_vs_reference_processor(golden, case, min_keep, mass):
    """The custom typical-sampling processor (utils/typical_sampling.py, model_v2.py:717-722) in the chain, against the
    reference's own class: same surviving set after it, same final probabilities."""
    g = golden("sampler_kat.npz")
    s_in = torch.from_numpy(g[f"{case}_in"])
    hist = g[f"{case}_hist"].tolist()
    tag = f"{case}_typ{int(mass * 100)}"
    kept = torch.isfinite(OG.typical_filter(OG.repetition_penalty(s_in, hist, 10.0), mass, min_keep))
    assert torch.equal(kept, torch.from_numpy(g[tag + "_kept"]))
    fin = OG.process_logits(s_in, hist, 10.0, 0.8, 30, 0.8, min_keep, typical_mass=mass)
    assert torch.allclose(torch.softmax(fin, -1), torch.from_numpy(g[tag + "_probs"]), atol=1e-6)


BEAM_TAGS = ["noeos", "mid", "mid2", "eos", "eos2", "lp1", "lpneg", "lp2noeos"]  # the last three: length_penalty 1.0 / -0.7 / 2.0


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("tag", BEAM_TAGS)
def test_beam_sample_vs_reference_scorer(golden, tag, batched):
    """3-beam beam-sample (the served default): oracle loop + scorer restatement vs a trace produced by
    the reference's own BeamSearchScorer / HF processors / model forward, replaying the recorded draws.  `batched`: the
    beams stepping through the trunk as one batch with an index_select-ed cache (the form the production-width GPU tests
    use), held to the same reference trace."""
    g = golden("gpt_beam.npz")
    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
    W["mel_head.bias"] = W["mel_head.bias"].clone()
    W["mel_head.bias"][8193] += float(g[f"{tag}_stop_bias"])
    orc = OG.GptOracle(W, cfg["layers"], cfg["heads"])
    fake, embeds, mask = orc.prepare_gpt_inputs(torch.from_numpy(g[f"{tag}_conds_latent"]), g[f"{tag}_text"])
    picks = g[f"{tag}_picks"]
    trace = []
    seq, score = OG.generate_beam_sample(orc, embeds, mask, int(g[f"{tag}_max_new"]), num_beams=3,
                                         sampler=lambda flat, step: picks[step - 1], trace=trace, batched=batched,
                                         length_penalty=float(g[f"{tag}_length_penalty"]) if f"{tag}_length_penalty" in g.files else 0.0)
    assert len(trace) == picks.shape[0]
    for t, ns, nt, ni in zip(trace, g[f"{tag}_next_scores"], g[f"{tag}_next_tokens"], g[f"{tag}_next_indices"]):
        assert t["next_tokens"] == nt.tolist() and t["next_indices"] == ni.tolist()
        assert np.allclose(t["next_scores"], ns, rtol=1e-4, atol=1e-3)
    assert trace[-1]["done"] == bool(g[f"{tag}_done"])
    assert seq == g[f"{tag}_sequence"].tolist()
    assert abs(score - float(g[f"{tag}_sequence_score"][0])) <= 1e-3 * max(1.0, abs(score))


BEAM_SEARCH_TAGS = [("noeos", 3), ("mid", 3), ("mid2", 3), ("eos", 3), ("lp1", 3), ("lp2", 3), ("nb2", 2), ("nb4", 4)]


@pytest.mark.parametrize("tag,nb", BEAM_SEARCH_TAGS)
def test_beam_search_without_sampling_vs_reference_scorer(golden, tag, nb):
    """`num_beams > 1, do_sample=False` (`_beam_search`'s top-k branch, generation_utils.py:3520-3524): the oracle picks its own
    candidates (no recorded draws to replay) and must walk the trace the reference's scorer + model forward produced."""
    g = golden("gpt_beam_search.npz")
    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
    W["mel_head.bias"] = W["mel_head.bias"].clone()
    W["mel_head.bias"][8193] += float(g[f"{tag}_stop_bias"])
    orc = OG.GptOracle(W, cfg["layers"], cfg["heads"])
    fake, embeds, mask = orc.prepare_gpt_inputs(torch.from_numpy(g[f"{tag}_conds_latent"]), g[f"{tag}_text"])
    trace = []
    seq, score = OG.generate_beam_search(orc, embeds, mask, int(g[f"{tag}_max_new"]), num_beams=nb, trace=trace, batched=True,
                                         length_penalty=float(g[f"{tag}_length_penalty"]))
    assert len(trace) == g[f"{tag}_picks"].shape[0]
    for t, pk, ns, nt, ni in zip(trace, g[f"{tag}_picks"], g[f"{tag}_next_scores"], g[f"{tag}_next_tokens"], g[f"{tag}_next_indices"]):
        assert t["picks"] == pk.tolist()
        assert t["next_tokens"] == nt.tolist() and t["next_indices"] == ni.tolist()
        assert np.allclose(t["next_scores"], ns, rtol=1e-4, atol=1e-3)
    assert trace[-1]["done"] == bool(g[f"{tag}_done"])
    assert seq == g[f"{tag}_sequence"].tolist()
    assert abs(score - float(g[f"{tag}_sequence_score"][0])) <= 1e-3 * max(1.0, abs(score))


def test_beam_replay_scores_the_reference_trace(golden):
    """`beam_replay` (a given run's per-step (token, source beam) pushed through the oracle -- what the bf16 production-width
    GPU test holds the device's free-running beams to) on the reference's own trace: every step's tokens were inside the
    processed support, and the accumulated increments are the reference's beam scores."""
    g = golden("gpt_beam.npz")
    tag = "noeos"
    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
    W["mel_head.bias"] = W["mel_head.bias"].clone()
    W["mel_head.bias"][8193] += float(g[f"{tag}_stop_bias"])
    orc = OG.GptOracle(W, cfg["layers"], cfg["heads"])
    fake, embeds, mask = orc.prepare_gpt_inputs(torch.from_numpy(g[f"{tag}_conds_latent"]), g[f"{tag}_text"])
    nt, ni, ns = g[f"{tag}_next_tokens"], g[f"{tag}_next_indices"], g[f"{tag}_next_scores"]
    rep = OG.beam_replay(orc, embeds, mask, nt, ni, keep_logits={1, len(nt)})
    score = np.array([0.0, -1e9, -1e9])
    for step, r in enumerate(rep):
        # (step 1 of the trace: beam 0 keeps two tokens, so the third draw of torch.multinomial is a zero-probability entry, score -inf)
        assert r["kept"] == np.isfinite(ns[step]).tolist(), step
        score = score[ni[step]] + np.array(r["inc"])
        assert np.allclose(score, ns[step], rtol=1e-4, atol=1e-3), step
    assert rep[0]["logits"].shape == (3, 8194) and "logits" not in rep[1]
