"""GPU parity: HIP GPT decode engine (through the C ABI) vs the reference fixtures and the oracle.

Bar (BASELINE.json north_star): greedy token ids bit-exact in fp32 mode.  Logits are compared
with a relative tolerance of 2e-4 of max|logit| (fp32 accumulation-order noise); the recorded
top-2 margins of the fixtures are >= 1e-2, i.e. far above that noise.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _tiny(g):
    import voice_tts_amd.weights as WR
    from oracle import gpt as OG

    cfg = WR.tiny_gpt_cfg(model_dim=int(g["model_dim"]), layers=int(g["layers"]), heads=int(g["heads"]))
    W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
    return cfg, W, OG.GptOracle(W, cfg["layers"], cfg["heads"])


@pytest.fixture(scope="module")
def tiny_f32(golden, dev):
    from voice_tts_amd.gpt_engine import GptEngine

    g = golden("gpt_tiny.npz")
    cfg, W, orc = _tiny(g)
    eng = GptEngine(cfg, dtype="f32", max_seq=256, max_batch=2, device=dev).load_state_dict(W)
    return g, cfg, W, orc, eng


@pytest.mark.parametrize("tag", ["plain", "padded"])
def test_tiny_greedy_ids_match_reference(tiny_f32, tag):
    g, cfg, W, orc, eng = tiny_f32
    embeds = torch.from_numpy(g[f"embeds_{tag}"])
    mask = g[f"mask_{tag}"]
    n_pad = int((mask == 0).sum())
    ref_ids = g[f"ids_{tag}"]
    n = len(ref_ids)
    eng.prefill(0, embeds, n_pad)
    l0 = eng.read_logits(0)
    ref_l = g[f"logits_{tag}"]
    assert np.abs(l0 - ref_l[0]).max() <= 2e-4 * np.abs(ref_l[0]).max()
    eng.decode(1, n, repetition_penalty=10.0)
    ids, fin = eng.read(0)
    assert ids.tolist() == ref_ids.tolist()
    assert g[f"margins_{tag}"].min() > 1e-2
    eng.prefill(0, embeds, n_pad)
    eng.decode(1, 1, repetition_penalty=10.0)
    l1 = eng.read_logits(0)
    assert np.abs(l1 - ref_l[1]).max() <= 2e-4 * np.abs(ref_l[1]).max()


def test_generate_surface_matches_reference_call(tiny_f32, dev):
    """store_mel_emb + generate(...) as inference_speech calls them (model_v2.py:698,724-729)."""
    g, cfg, W, orc, eng = tiny_f32
    embeds = torch.from_numpy(g["embeds_padded"]).unsqueeze(0).to(dev)
    mask = torch.from_numpy(g["mask_padded"]).unsqueeze(0).to(dev)
    P = mask.shape[1]
    fake = torch.ones(1, P, dtype=torch.long, device=dev)
    fake[0, -1] = 8192
    eng.store_mel_emb(embeds)
    out = eng.generate(fake, bos_token_id=8192, pad_token_id=8193, eos_token_id=8193, attention_mask=mask,
                       max_length=P + 40, num_return_sequences=1, do_sample=True, top_p=0.8, top_k=1,
                       temperature=0.8, num_beams=1, repetition_penalty=10.0, length_penalty=0.0, sync_every=16)
    assert out.dtype == torch.long and out.shape == (1, P + 40)
    assert out[0, :P].tolist() == fake[0].tolist()
    assert out[0, P:].tolist() == g["ids_padded"].tolist()
    with pytest.raises(NotImplementedError):
        eng.generate(fake, attention_mask=mask, max_length=P + 4, num_beams=3)  # this engine was built with max_batch=2
    # sampling through the same surface (served defaults except num_beams): ids stay in range, length honoured
    out2 = eng.generate(fake, attention_mask=mask, max_length=P + 12, do_sample=True, top_p=0.8, top_k=30, temperature=0.8,
                        num_beams=1, repetition_penalty=10.0, seed=7)
    assert out2.shape[1] <= P + 12 and int(out2[0, P:].max()) < 8194


def test_teacher_forced_logits_vs_oracle(tiny_f32):
    from oracle import gpt as OG

    g, cfg, W, orc, eng = tiny_f32
    embeds = torch.from_numpy(g["embeds_plain"])
    mask = torch.from_numpy(g["mask_plain"])
    forced = [7, 8193 - 5, 4000, 17, 17, 256, 8191, 3]
    ids, margins, logits = OG.generate_greedy(orc, embeds, mask, len(forced), return_logits=True, forced=forced)
    eng.prefill(0, embeds, 0)
    for k, tok in enumerate(forced):
        got = eng.read_logits(0)
        ref = logits[k].numpy()
        assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max(), k
        eng.force_next(0, tok)
        eng.decode(1, 1, repetition_penalty=10.0)
    out, _ = eng.read(0)
    assert out.tolist() == forced


def test_two_slots_decode_together(tiny_f32):
    """Batch 2 with different prompts/lengths == each sequence alone (segment batching, row N3)."""
    g, cfg, W, orc, eng = tiny_f32
    eng.prefill(0, torch.from_numpy(g["embeds_plain"]), 0)
    eng.prefill(1, torch.from_numpy(g["embeds_padded"]), 3)
    eng.decode(2, 40, repetition_penalty=10.0)
    a, _ = eng.read(0)
    b, _ = eng.read(1)
    assert a.tolist() == g["ids_plain"].tolist()
    assert b.tolist() == g["ids_padded"].tolist()


def test_stop_token_and_suppress(tiny_f32):
    from oracle import gpt as OG

    g, cfg, W, orc, eng = tiny_f32
    embeds = torch.from_numpy(g["embeds_plain"])
    mask = torch.from_numpy(g["mask_plain"])
    eng.prefill(0, embeds, 0)
    eng.force_next(0, 8193)
    eng.decode(1, 5, repetition_penalty=10.0)
    ids, fin = eng.read(0)
    assert fin and ids.tolist() == [8193]  # finished rows keep emitting pad; read() trims at the first stop
    ref, _ = OG.generate_greedy(orc, embeds, mask, 12, suppress_stop=True)
    eng.prefill(0, embeds, 0)
    eng.decode(1, 12, repetition_penalty=10.0, suppress_stop=True)
    ids, fin = eng.read(0)
    assert ids.tolist() == ref and not fin


def test_sampling_distribution_vs_oracle(tiny_f32):
    """Config 3: do_sample with repetition_penalty 10, temperature 0.8, top_k 30, top_p 0.8.
    RNG cannot match torch's CPU stream, so parity is on the processed probability vector
    (max-abs <= 1e-5) plus a frequency check of the draws."""
    from oracle import gpt as OG

    g, cfg, W, orc, eng = tiny_f32
    embeds = torch.from_numpy(g["embeds_plain"])
    mask = torch.from_numpy(g["mask_plain"])
    P = len(mask)
    forced = [11, 4097, 256]
    eng.prefill(0, embeds, 0)
    for tok in forced:
        eng.force_next(0, tok)
        eng.decode(1, 1, repetition_penalty=10.0)
    logits = torch.from_numpy(eng.read_logits(0))
    hist = [1] * (P - 1) + [8192] + forced
    for (T, k, p) in [(0.8, 30, 0.8), (1.3, 5, 0.5), (0.8, 128, 1.0), (0.7, 1, 0.9)]:
        ref = torch.softmax(OG.process_logits(logits, hist, 10.0, T, k, p, 1), -1).numpy()
        counts = {}
        n_draw = 160 if (T, k, p) == (0.8, 30, 0.8) else 3
        for seed in range(n_draw):
            eng.prefill(0, embeds, 0)
            for tok in forced:
                eng.force_next(0, tok)
                eng.decode(1, 1, repetition_penalty=10.0)
            eng.decode(1, 1, repetition_penalty=10.0, temperature=T, top_k=k, top_p=p, do_sample=True, seed=1000 + seed)
            ids, _ = eng.read(0)
            tok = int(ids[len(forced)])
            counts[tok] = counts.get(tok, 0) + 1
            assert ref[tok] > 0, (tok, T, k, p)  # every draw lies in the support the reference would sample from
        probs = eng.read_probs(0)
        assert np.abs(probs - ref).max() <= 1e-5, (T, k, p, np.abs(probs - ref).max())
        assert (probs > 0).sum() == (ref > 0).sum()
        if n_draw >= 100:
            top = int(ref.argmax())
            f = counts.get(top, 0) / n_draw
            sigma = (ref[top] * (1 - ref[top]) / n_draw) ** 0.5
            assert abs(f - ref[top]) <= 4.5 * sigma + 1e-9, (f, ref[top])
            assert len(counts) > 1 or ref[top] > 0.97


def test_top_k_when_one_thread_holds_the_large_scores(dev):
    """`topk_sorted_1024` bounds the k-th largest score by the k-th largest per-thread maximum; a thread owns the tokens t, t + 1024,
    ... -- with the 30 largest logits placed on 4 threads (and with every logit equal) the candidate pool degenerates and the
    kernel must take its exact four-pass select.  Probabilities against the oracle's processors either way."""
    import voice_tts_amd.weights as WR
    from oracle import gpt as OG
    from voice_tts_amd.gpt_engine import GptEngine

    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    W = WR.make_gpt_weights(cfg, seed=5)
    W["mel_head.weight"] = torch.zeros_like(W["mel_head.weight"])  # logits = the head's bias, whatever the hidden state
    g = torch.Generator().manual_seed(6)
    emb = torch.randn(20, 128, generator=g) * 0.5
    for case in ("clustered", "flat"):
        bias = torch.randn(8194, generator=g) * 0.1 if case == "clustered" else torch.zeros(8194)
        if case == "clustered":
            big = [t + 1024 * i for t in (5, 6, 7, 700) for i in range(8)]  # 32 large values on four threads
            bias[big] = 8.0 + torch.rand(len(big), generator=g)
        W["mel_head.bias"] = bias
        eng = GptEngine(cfg, dtype="f32", max_seq=64, max_batch=1, device=dev).load_state_dict(W)
        eng.prefill(0, emb, 0)
        eng.decode(1, 1, repetition_penalty=1.0, temperature=0.9, top_k=30, top_p=1.0, do_sample=True, seed=3)
        ids, _ = eng.read(0)
        probs = eng.read_probs(0)
        if case == "clustered":
            ref = torch.softmax(OG.process_logits(bias.clone(), [], 1.0, 0.9, 30, 1.0, 1), -1).numpy()
            assert np.abs(probs - ref).max() <= 1e-5 and (probs > 0).sum() == 30
            assert ref[int(ids[0])] > 0
        else:  # every token ties with the k-th: HF keeps them all; the device keeps its 128-entry survivor list, uniformly
            kept = probs[probs > 0]
            assert len(kept) == 128 and np.allclose(kept, 1.0 / 128, rtol=1e-5) and probs[int(ids[0])] > 0


def test_top_k_ties_at_the_kth_value_keep_the_larger_scores_and_are_deterministic(dev):
    """3 scores strictly above a plateau of 600 equal ones, top_k = 30: the k-th largest value IS the plateau.  HF keeps every
    tie; the device's survivor list holds 128, so it keeps the 3 larger scores and fills up with plateau tokens LOWEST ID FIRST
    -- the same set on every run (the pool overflows, so this takes the exact-select path of `topk_sorted_1024`)."""
    import voice_tts_amd.weights as WR
    from voice_tts_amd.gpt_engine import GptEngine

    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    W = WR.make_gpt_weights(cfg, seed=5)
    W["mel_head.weight"] = torch.zeros_like(W["mel_head.weight"])
    g = torch.Generator().manual_seed(8)
    bias = torch.randn(8194, generator=g) * 0.1
    plateau = torch.randperm(8194, generator=g)[:603]
    big, plateau = plateau[:3].tolist(), sorted(plateau[3:].tolist())
    bias[plateau] = 4.0
    bias[big] = torch.tensor([6.0, 5.5, 5.0])
    W["mel_head.bias"] = bias
    emb = torch.randn(20, 128, generator=g) * 0.5
    eng = GptEngine(cfg, dtype="f32", max_seq=64, max_batch=1, device=dev).load_state_dict(W)
    supports = []
    for seed in range(4):
        eng.prefill(0, emb, 0)
        eng.decode(1, 1, repetition_penalty=1.0, temperature=1.0, top_k=30, top_p=1.0, do_sample=True, seed=seed)
        probs = eng.read_probs(0)
        supports.append(np.nonzero(probs > 0)[0].tolist())
        ids, _ = eng.read(0)
        assert probs[int(ids[0])] > 0
    assert all(s == supports[0] for s in supports)
    assert len(supports[0]) == 128 and set(big) <= set(supports[0])
    assert sorted(set(supports[0]) - set(big)) == plateau[:125]  # lowest ids first
    p = eng.read_probs(0)
    z = np.exp(6.0) + np.exp(5.5) + np.exp(5.0) + 125 * np.exp(4.0)
    assert np.allclose(p[big], np.exp([6.0, 5.5, 5.0]) / z, rtol=1e-5) and np.allclose(p[plateau[:125]], np.exp(4.0) / z, rtol=1e-5)


def test_sampling_with_unbounded_top_k_vs_oracle(tiny_f32):
    """G8 corners generate() reaches through infer(top_k=..., top_p=...): top_k = 0 (off), top_k beyond the 128-entry survivor
    list, top_p = 1.0, a top_p that leaves only min_tokens_to_keep.  The device's processed probability vector equals the
    oracle's (itself pinned to the HF warpers for these very settings, tests/golden/sampler_kat.npz `*_wide_*`) <= 1e-5 with the
    same support, every draw lies in that support, and the draw frequencies follow it."""
    from oracle import gpt as OG

    g, cfg, W, orc, eng = tiny_f32
    embeds = torch.from_numpy(g["embeds_plain"])
    P = embeds.shape[0] + 1
    forced = [11, 4097, 256]

    def replay():
        eng.prefill(0, embeds, 0)
        for tok in forced:
            eng.force_next(0, tok)
            eng.decode(1, 1, repetition_penalty=10.0)

    replay()
    logits = torch.from_numpy(eng.read_logits(0))
    hist = [1] * (P - 1) + [8192] + forced
    for (T, k, p) in [(0.8, 0, 0.8), (0.9, 500, 0.95), (0.8, 129, 0.6), (1.3, 0, 1.0), (0.8, 0, 0.02), (1.0, 4000, 0.999), (0.8, 9000, 0.9)]:
        ref = torch.softmax(OG.process_logits(logits, hist, 10.0, T, k if k < 8194 else 0, p, 1), -1).numpy()
        n_draw = 200 if (T, k, p) == (0.8, 0, 0.8) else 3
        counts = {}
        for seed in range(n_draw):
            replay()
            eng.decode(1, 1, repetition_penalty=10.0, temperature=T, top_k=k, top_p=p, do_sample=True, seed=500 + seed)
            tok = int(eng.read(0)[0][len(forced)])
            counts[tok] = counts.get(tok, 0) + 1
            assert ref[tok] > 0, (tok, T, k, p)
        probs = eng.read_probs(0)
        assert (probs > 0).sum() == (ref > 0).sum(), (T, k, p, (probs > 0).sum(), (ref > 0).sum())
        assert np.abs(probs - ref).max() <= 1e-5, (T, k, p)
        if n_draw >= 100:
            order = np.argsort(-ref)[:3]
            for t in order:  # three most likely tokens: observed frequency within 4 sigma of p
                ph, pr = counts.get(int(t), 0) / n_draw, float(ref[t])
                assert abs(ph - pr) <= 4 * np.sqrt(pr * (1 - pr) / n_draw) + 1e-3, (int(t), ph, pr)
    # the reference-shaped surface takes the same settings (model_v2.py:724-729)
    dev = torch.device("cuda:0")
    eng.store_mel_emb(embeds.unsqueeze(0).to(dev))
    fk = torch.ones(1, P, dtype=torch.long, device=dev)
    fk[0, -1] = 8192
    out = eng.generate(fk, bos_token_id=8192, pad_token_id=8193, eos_token_id=8193, max_length=P + 12, do_sample=True, top_p=0.9, top_k=0,
                       temperature=0.8, num_beams=1, repetition_penalty=10.0, seed=4)
    assert out.shape[0] == 1 and P < out.shape[1] <= P + 12
    # num_return_sequences (generation_utils.py:2128-2135): independent draws of one prompt, one row each; the first row is the
    # single-sequence draw of the same seed (slot 0's stream), the rows differ from each other
    if eng.max_batch >= 2:
        two = eng.generate(fk, bos_token_id=8192, pad_token_id=8193, eos_token_id=8193, max_length=P + 12, do_sample=True, top_p=0.9, top_k=0,
                           temperature=0.8, num_beams=1, repetition_penalty=10.0, seed=4, num_return_sequences=2)
        assert two.shape[0] == 2 and two[0, : out.shape[1]].tolist() == out[0].tolist() and two[0].tolist() != two[1].tolist()


def test_typical_sampling_on_device_vs_oracle(tiny_f32):
    """`inference_speech(typical_sampling=True, typical_mass=m)` (model_v2.py:717-722): the TypicalLogitsWarper runs inside the
    sampler kernel between the repetition penalty and the warpers; the processed probability vector matches the oracle
    (itself pinned to the reference's class), for sampling and for the greedy argmax."""
    from oracle import gpt as OG

    g, cfg, W, orc, eng = tiny_f32
    embeds = torch.from_numpy(g["embeds_plain"])
    P = len(g["mask_plain"])
    forced = [11, 4097, 256]

    def replay():
        eng.prefill(0, embeds, 0)
        for tok in forced:
            eng.force_next(0, tok)
            eng.decode(1, 1, repetition_penalty=10.0)

    replay()
    logits = torch.from_numpy(eng.read_logits(0))
    hist = [1] * (P - 1) + [8192] + forced
    for mass in (0.9, 0.5, 0.2):
        ref = torch.softmax(OG.process_logits(logits, hist, 10.0, 0.8, 30, 0.8, 1, typical_mass=mass), -1).numpy()
        replay()
        eng.decode(1, 1, repetition_penalty=10.0, temperature=0.8, top_k=30, top_p=0.8, do_sample=True, seed=5, typical_mass=mass)
        probs = eng.read_probs(0)
        assert (probs > 0).sum() == (ref > 0).sum(), (mass, (probs > 0).sum(), (ref > 0).sum())
        assert np.abs(probs - ref).max() <= 1e-5, (mass, np.abs(probs - ref).max())
        assert ref[int(eng.read(0)[0][len(forced)])] > 0
        # greedy through the same processor: argmax of the typical-filtered scores
        replay()
        eng.decode(1, 1, repetition_penalty=10.0, typical_mass=mass)
        want = int(torch.argmax(OG.process_logits(logits, hist, 10.0, typical_mass=mass)))
        assert int(eng.read(0)[0][len(forced)]) == want


def test_typical_sampling_with_beams_runs(tiny_f32):
    g, cfg, W, orc, eng3 = tiny_f32
    from voice_tts_amd.gpt_engine import GptEngine

    eng = GptEngine(cfg, dtype="f32", max_seq=256, max_batch=3, device=eng3.device).load_state_dict(W)
    eng.prefill(0, torch.from_numpy(g["embeds_plain"]), 0)
    eng.beam_begin(3)
    eng.beam_decode(12, typical_mass=0.9, suppress_stop=True, seed=3)
    ids = eng.beam_read(12)[0]
    assert len(ids) == 12 and int(ids.max()) < 8194 and int(ids.min()) >= 0


def test_latent_pass_vs_reference(tiny_f32, dev):
    g, cfg, W, orc, eng = tiny_f32
    conds = torch.from_numpy(g["conds_latent"])
    text = torch.from_numpy(g["text_plain"]).long()
    t = torch.cat((torch.tensor([0]), text, torch.tensor([1])))
    temb = W["text_embedding.weight"][t] + W["text_pos_embedding.emb.weight"][: t.numel()]
    prefix = torch.cat((conds, temb), 0)
    lat = eng.latent(prefix, torch.from_numpy(g["latent_codes"])).cpu()
    ref = torch.from_numpy(g["latent"])
    assert lat.shape == ref.shape
    assert (lat - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


def test_bf16_mode_against_bf16_rounded_oracle(golden, dev):
    """Throughput mode: weights/KV in bf16, fp32 accumulate.  Checker = the oracle run on the
    SAME bf16-rounded weights (isolates kernel correctness from quantisation)."""
    from oracle import gpt as OG
    from voice_tts_amd.gpt_engine import GptEngine

    g = golden("gpt_tiny.npz")
    cfg, W, _ = _tiny(g)
    mats = ("c_attn.weight", "c_proj.weight", "c_fc.weight", "mel_head.weight")
    Wq = {k: (v.to(torch.bfloat16).to(torch.float32) if k.endswith(mats) else v) for k, v in W.items()}
    orc = OG.GptOracle(Wq, cfg["layers"], cfg["heads"])
    eng = GptEngine(cfg, dtype="bf16", max_seq=256, max_batch=1, device=dev).load_state_dict(W)
    embeds = torch.from_numpy(g["embeds_plain"])
    mask = torch.from_numpy(g["mask_plain"])
    ids, margins, logits = OG.generate_greedy(orc, embeds, mask, 24, return_logits=True)
    eng.prefill(0, embeds, 0)
    got0 = eng.read_logits(0)
    # KV is rounded to bf16 on the device but not in the checker: tolerance 1e-2 of max|logit|
    assert np.abs(got0 - logits[0].numpy()).max() <= 1e-2 * np.abs(logits[0].numpy()).max()
    eng.decode(1, 24, repetition_penalty=10.0)
    out, _ = eng.read(0)
    agree = np.mean(np.array(out.tolist()) == np.array(ids))
    assert agree >= 0.5, agree  # sequences may fork at a near-tie; the first tokens must agree
    assert out.tolist()[:4] == ids[:4]


def test_production_width_layer_vs_reference(golden, dev):
    import voice_tts_amd.weights as WR
    from oracle import gpt as OG
    from voice_tts_amd.gpt_engine import GptEngine

    g = golden("gpt_prod_layer.npz")
    cfg = dict(WR.GPT_CFG)
    cfg.update(layers=1, max_text_tokens=40, max_mel_tokens=80, number_text_tokens=200)
    W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
    orc = OG.GptOracle(W, 1, cfg["heads"])
    fake, embeds, mask = orc.prepare_gpt_inputs(torch.from_numpy(g["conds_latent"]), g["text"])
    eng = GptEngine(cfg, dtype="f32", max_seq=128, max_batch=1, device=dev).load_state_dict(W)
    eng.prefill(0, embeds, 0)
    l0 = eng.read_logits(0)
    assert np.abs(l0 - g["logits_first"]).max() <= 2e-4 * np.abs(g["logits_first"]).max()
    eng.decode(1, 6, repetition_penalty=10.0)
    ids, _ = eng.read(0)
    assert ids.tolist() == g["ids"].tolist()


def test_errors(tiny_f32, dev):
    from voice_tts_amd._lib import IxttsError
    from voice_tts_amd.gpt_engine import GptEngine

    g, cfg, W, orc, eng = tiny_f32
    with pytest.raises(IxttsError):
        eng.prefill(5, torch.zeros(4, cfg["model_dim"]), 0)  # slot out of range
    with pytest.raises(IxttsError):
        eng.prefill(0, torch.zeros(300, cfg["model_dim"]), 0)  # longer than max_seq
    e2 = GptEngine(cfg, dtype="f32", max_seq=64, max_batch=1, device=dev)
    with pytest.raises(IxttsError):
        e2.load_state_dict({k: v for k, v in W.items() if k != "final_norm.bias"})  # finalize reports the gap
    with pytest.raises(IxttsError):
        e2.decode(1, 1)  # not finalized


@pytest.fixture(scope="module")
def beam_engines(golden, dev):
    """One fp32 engine per stop-token bias used by the beam fixtures (max_batch 3 = the served num_beams)."""
    import voice_tts_amd.weights as WR
    from oracle import gpt as OG
    from voice_tts_amd.gpt_engine import GptEngine

    g = golden("gpt_beam.npz")
    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    out = {}
    for tag in ("noeos", "mid", "mid2", "eos", "eos2", "lp1", "lpneg", "lp2noeos"):
        W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
        W["mel_head.bias"] = W["mel_head.bias"].clone()
        W["mel_head.bias"][8193] += float(g[f"{tag}_stop_bias"])
        orc = OG.GptOracle(W, cfg["layers"], cfg["heads"])
        eng = GptEngine(cfg, dtype="f32", max_seq=128, max_batch=3, device=dev).load_state_dict(W)
        out[tag] = (orc, eng)
    return g, out


@pytest.mark.parametrize("tag", ["noeos", "mid", "mid2", "eos", "eos2", "lp1", "lpneg", "lp2noeos"])
def test_beam_sample_replays_reference_trace(beam_engines, tag):
    """Served default (num_beams=3, do_sample): device processors + BeamSearchScorer bookkeeping + KV reorder,
    replaying the draws recorded from the reference's own scorer run (tests/golden/gpt_beam.npz)."""
    g, engines = beam_engines
    orc, eng = engines[tag]
    fake, embeds, mask = orc.prepare_gpt_inputs(torch.from_numpy(g[f"{tag}_conds_latent"]), g[f"{tag}_text"])
    picks = g[f"{tag}_picks"]
    max_new = int(g[f"{tag}_max_new"])
    lp = float(g[f"{tag}_length_penalty"]) if f"{tag}_length_penalty" in g.files else 0.0  # lp1 / lpneg / lp2noeos: 1.0 / -0.7 / 2.0
    eng.prefill(0, embeds, 0)
    eng.beam_begin(3)
    for step in range(picks.shape[0]):
        eng.beam_force(picks[step])
        eng.beam_decode(1, repetition_penalty=10.0, temperature=0.8, top_k=30, top_p=0.8, length_penalty=lp)
        ids, done, score, bs, lt, src = eng.beam_read(max_new)
        assert lt.tolist() == g[f"{tag}_next_tokens"][step].tolist(), step
        assert src.tolist() == g[f"{tag}_next_indices"][step].tolist(), step
        assert np.allclose(bs, g[f"{tag}_next_scores"][step], rtol=1e-4, atol=2e-3), step
    assert done == bool(g[f"{tag}_done"])
    assert ids.tolist() == g[f"{tag}_sequence"].tolist()
    assert abs(score - float(g[f"{tag}_sequence_score"][0])) <= 2e-3 * max(1.0, abs(score))
    # once done, further steps change nothing (HF leaves the loop)
    if done:
        eng.beam_decode(3, repetition_penalty=10.0, length_penalty=lp)
        ids2, done2 = eng.beam_read(max_new)[:2]
        assert done2 and ids2.tolist() == ids.tolist()


@pytest.mark.parametrize("tag,nb", [("noeos", 3), ("mid", 3), ("mid2", 3), ("eos", 3), ("lp1", 3), ("lp2", 3), ("nb2", 2), ("nb4", 4)])
def test_beam_search_without_sampling_walks_the_reference_trace(golden, dev, tag, nb):
    """`num_beams > 1, do_sample=False` (`_beam_search`'s topk branch, generation_utils.py:3520-3524): the device picks its own
    candidates -- the joint top 2 * num_beams, no warpers -- and must walk, step by step, the trace the reference's scorer + model
    forward produced (tests/golden/gpt_beam_search.npz); then the same through `generate()`."""
    import voice_tts_amd.weights as WR
    from oracle import gpt as OG
    from voice_tts_amd.gpt_engine import GptEngine

    g = golden("gpt_beam_search.npz")
    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
    W["mel_head.bias"] = W["mel_head.bias"].clone()
    W["mel_head.bias"][8193] += float(g[f"{tag}_stop_bias"])
    orc = OG.GptOracle(W, cfg["layers"], cfg["heads"])
    eng = GptEngine(cfg, dtype="f32", max_seq=128, max_batch=nb, device=dev).load_state_dict(W)
    fake, embeds, mask = orc.prepare_gpt_inputs(torch.from_numpy(g[f"{tag}_conds_latent"]), g[f"{tag}_text"])
    max_new, lp = int(g[f"{tag}_max_new"]), float(g[f"{tag}_length_penalty"])
    eng.prefill(0, embeds, 0)
    eng.beam_begin(nb)
    n_steps = g[f"{tag}_picks"].shape[0]
    for step in range(n_steps):
        eng.beam_decode(1, repetition_penalty=10.0, length_penalty=lp, do_sample=False)
        ids, done, score, bs, lt, src = eng.beam_read(max_new)
        assert lt.tolist() == g[f"{tag}_next_tokens"][step].tolist(), step
        assert src.tolist() == g[f"{tag}_next_indices"][step].tolist(), step
        assert np.allclose(bs, g[f"{tag}_next_scores"][step], rtol=1e-4, atol=2e-3), step
    assert done == bool(g[f"{tag}_done"])
    assert ids.tolist() == g[f"{tag}_sequence"].tolist()
    assert abs(score - float(g[f"{tag}_sequence_score"][0])) <= 2e-3 * max(1.0, abs(score))
    # the generate() surface (model_v2.py:724-729 with do_sample=False, num_beams=nb)
    eng.store_mel_emb(embeds.unsqueeze(0))
    out = eng.generate(torch.ones(1, embeds.shape[0] + 1, dtype=torch.long), max_length=embeds.shape[0] + 1 + max_new, do_sample=False, num_beams=nb,
                       repetition_penalty=10.0, length_penalty=lp, temperature=0.8, top_k=30, top_p=0.8)  # warper settings are ignored
    assert out[0, embeds.shape[0] + 1:].tolist() == g[f"{tag}_sequence"].tolist()


def test_beam_sample_free_running_and_generate_surface(beam_engines, dev):
    from oracle import gpt as OG

    g, engines = beam_engines
    orc, eng = engines["noeos"]
    fake, embeds, mask = orc.prepare_gpt_inputs(torch.from_numpy(g["noeos_conds_latent"]), g["noeos_text"])
    P = len(mask)
    # step 1: beam_scores = [0,-1e9,-1e9] -> the three new beams are distinct tokens of beam 0's filtered support
    logits0, _ = orc.prefill(embeds, mask)
    support = OG.process_logits(torch.log_softmax(logits0, -1), [1] * (P - 1) + [8192], 10.0, 0.8, 30, 0.8, min_keep=2)
    allowed = set(torch.nonzero(~torch.isinf(support)).flatten().tolist())
    firsts = set()
    for seed in range(12):
        eng.prefill(0, embeds, 0)
        eng.beam_begin(3)
        eng.beam_decode(1, seed=seed)
        ids, done, score, bs, lt, src = eng.beam_read(24)
        assert src.tolist() == [0, 0, 0] or not (bs > -1e8).all()
        live = [int(t) for t, s in zip(lt, bs) if s > -1e8]
        assert len(set(live)) == len(live) and set(live) <= allowed
        firsts.update(live)
    assert len(firsts) > 1
    # the reference-shaped call with the served defaults (model_v2.py:724-729, infer_v2.py:598-606)
    eng.store_mel_emb(embeds.unsqueeze(0).to(dev))
    fk = torch.ones(1, P, dtype=torch.long, device=dev)
    fk[0, -1] = 8192
    out = eng.generate(fk, bos_token_id=8192, pad_token_id=8193, eos_token_id=8193, attention_mask=mask.unsqueeze(0).to(dev),
                       max_length=P + 20, num_return_sequences=1, do_sample=True, top_p=0.8, top_k=30, temperature=0.8,
                       num_beams=3, repetition_penalty=10.0, length_penalty=0.0, seed=3)
    assert out.shape[0] == 1 and P < out.shape[1] <= P + 20 and int(out[0, P:].max()) < 8194


def test_beam_kv_reorder_moves_only_unshared_rows_same_result(beam_engines, dev, monkeypatch):
    """`_reorder_cache` (model_v2.py:199-212) index_selects every K/V row; the device moves only the rows two slots do not
    already share (gpt_beam.hip: beam_reorder_kv_kernel).  Same free-running 3-beam decode on an engine that moves every row
    (IXTTS_BEAM_REORDER=full): tokens, source beams and beam scores must agree bit for bit, step by step."""
    import voice_tts_amd.weights as WR
    from voice_tts_amd.gpt_engine import GptEngine

    g, engines = beam_engines
    orc, eng = engines["noeos"]
    fake, embeds, mask = orc.prepare_gpt_inputs(torch.from_numpy(g["noeos_conds_latent"]), g["noeos_text"])
    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
    W["mel_head.bias"] = W["mel_head.bias"].clone()
    W["mel_head.bias"][8193] += float(g["noeos_stop_bias"])
    monkeypatch.setenv("IXTTS_BEAM_REORDER", "full")
    full = GptEngine(cfg, dtype="f32", max_seq=128, max_batch=3, device=dev).load_state_dict(W)
    monkeypatch.delenv("IXTTS_BEAM_REORDER")
    n_src_patterns = set()
    for seed in (1, 2, 3):
        for e in (eng, full):
            e.prefill(0, embeds, 0)
            e.beam_begin(3)
        for step in range(40):
            outs = []
            for e in (eng, full):
                # (temperature 3: flat enough that the beams keep swapping ancestors)
                e.beam_decode(1, repetition_penalty=10.0, temperature=3.0, top_k=30, top_p=0.95, suppress_stop=True, seed=seed)
                outs.append(e.beam_read(64))
            (ids_a, done_a, sc_a, bs_a, lt_a, src_a), (ids_b, done_b, sc_b, bs_b, lt_b, src_b) = outs
            assert lt_a.tolist() == lt_b.tolist() and src_a.tolist() == src_b.tolist(), (seed, step)
            assert np.array_equal(bs_a, bs_b), (seed, step, bs_a, bs_b)
            n_src_patterns.add(tuple(src_a.tolist()))
        assert ids_a.tolist() == ids_b.tolist()
    assert len(n_src_patterns) >= 4, n_src_patterns  # identity, collapses onto one beam, swaps


def test_beam_joint_draw_frequencies_match_multinomial_without_replacement(beam_engines):
    """`_beam_search` draws 2 * num_beams flat indices with `torch.multinomial(probs, 6)` -- WITHOUT replacement
    (transformers_generation_utils.py:3473-3530).  The device draws them jointly by Gumbel-top-k; here its kept beams are
    counted over 800 seeds on an 8-token support (top_k 8, top_p 1) and held to the exact law of sequential sampling without
    replacement (all 8P6 ordered draws enumerated): the 3 kept beams are the 3 best-scoring of the 6 drawn."""
    import itertools

    from oracle import gpt as OG

    g, engines = beam_engines
    orc, eng = engines["noeos"]
    fake, embeds, mask = orc.prepare_gpt_inputs(torch.from_numpy(g["noeos_conds_latent"]), g["noeos_text"])
    P = len(mask)
    logits0, _ = orc.prefill(embeds, mask)
    TEMP = 25.0  # the synthetic head is sharp (x50): a high temperature spreads the mass over the 8 survivors, so many kept sets occur
    sc = OG.process_logits(torch.log_softmax(logits0, -1), [1] * (P - 1) + [8192], 10.0, TEMP, 8, 1.0, min_keep=2)
    toks = torch.nonzero(torch.isfinite(sc)).flatten().tolist()
    assert len(toks) == 8
    p = torch.softmax(sc[toks].double(), -1).tolist()
    score = {t: float(sc[t]) for t in toks}
    # exact law of the kept set: enumerate the ordered draws of 6 out of 8
    law = {}
    for seq in itertools.permutations(range(8), 6):
        pr, rest = 1.0, 1.0
        for i in seq:
            pr *= p[i] / rest
            rest -= p[i]
        kept = frozenset(sorted((toks[i] for i in seq), key=lambda t: -score[t])[:3])
        law[kept] = law.get(kept, 0.0) + pr
    assert abs(sum(law.values()) - 1.0) < 1e-9
    N = 800
    seen = {}
    for seed in range(N):
        eng.prefill(0, embeds, 0)
        eng.beam_begin(3)
        eng.beam_decode(1, repetition_penalty=10.0, temperature=TEMP, top_k=8, top_p=1.0, seed=10_000 + seed)
        ids, done, s_, bs, lt, src = eng.beam_read(24)
        kept = frozenset(int(t) for t in lt)
        assert len(kept) == 3 and kept <= set(toks) and src.tolist() == [0, 0, 0]
        seen[kept] = seen.get(kept, 0) + 1
    assert set(seen) <= set(law)
    # every outcome with a non-negligible probability within 4 sigma; and a chi-square-style total over the common ones
    chi2, dof = 0.0, 0
    for kept, pr in law.items():
        obs = seen.get(kept, 0)
        assert abs(obs / N - pr) <= 4 * np.sqrt(pr * (1 - pr) / N) + 2e-3, (sorted(kept), obs / N, pr)
        if pr * N >= 5:
            chi2 += (obs - pr * N) ** 2 / (pr * N)
            dof += 1
    assert dof >= 8 and chi2 <= 2.5 * dof + 10, (chi2, dof, max(law.values()))


def test_legacy_attention_path(golden, dev, monkeypatch):
    """IXTTS_ATTN=legacy selects the any-length one-workgroup-per-head attention kernel (the fallback for contexts beyond the
    largest split-S bucket); the default split-S path is what every other test runs."""
    from voice_tts_amd.gpt_engine import GptEngine

    g = golden("gpt_tiny.npz")
    cfg, W, orc = _tiny(g)
    monkeypatch.setenv("IXTTS_ATTN", "legacy")
    eng = GptEngine(cfg, dtype="f32", max_seq=256, max_batch=2, device=dev).load_state_dict(W)
    eng.prefill(0, torch.from_numpy(g["embeds_plain"]), 0)
    eng.prefill(1, torch.from_numpy(g["embeds_padded"]), 3)
    eng.decode(2, 40, repetition_penalty=10.0)
    assert eng.read(0)[0].tolist() == g["ids_plain"].tolist()
    assert eng.read(1)[0].tolist() == g["ids_padded"].tolist()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_split_attention_across_context_buckets(golden, dev, monkeypatch, dtype):
    """Greedy decode from a 200-row prompt through 700 steps: the split-S attention switches instantiation at every 256-key
    bucket boundary (the host picks it per graph launch).  Same tokens as the any-length one-workgroup-per-head kernel, and
    (fp32) as the CPU oracle over the first 330 steps (two boundaries)."""
    from oracle import gpt as OG
    from voice_tts_amd.gpt_engine import GptEngine

    import voice_tts_amd.weights as WR

    g = golden("gpt_tiny.npz")
    # the fixture's model has 83 mel positions; this one has room for the whole run so the oracle can follow it
    cfg = WR.tiny_gpt_cfg(model_dim=int(g["model_dim"]), layers=int(g["layers"]), heads=int(g["heads"]), max_mel_tokens=800)
    W = WR.make_gpt_weights(cfg, seed=int(g["seed"]), head_scale=50.0)
    orc = OG.GptOracle(W, cfg["layers"], cfg["heads"])
    gen = torch.Generator().manual_seed(77)
    emb = torch.randn(200, cfg["model_dim"], generator=gen) * 0.5
    n = 700

    def run(legacy):
        if legacy:
            monkeypatch.setenv("IXTTS_ATTN", "legacy")
        else:
            monkeypatch.delenv("IXTTS_ATTN", raising=False)
        eng = GptEngine(cfg, dtype=dtype, max_seq=1024, max_batch=2, device=dev).load_state_dict(W)
        eng.prefill(0, emb, 0)
        eng.prefill(1, emb[:150], 0)
        eng.decode(2, n, repetition_penalty=10.0, suppress_stop=True)
        return eng.read(0)[0][:n].tolist(), eng.read(1)[0][:n].tolist()

    a0, a1 = run(False)
    b0, b1 = run(True)
    assert a0 == b0 and a1 == b1
    if dtype == "f32":
        mask = torch.ones(201, dtype=torch.long)
        ref, margins = OG.generate_greedy(orc, emb, mask, 330, suppress_stop=True)
        ref, margins = list(ref), np.asarray(margins)
        close = np.nonzero(margins < 1e-3)[0]  # a near-tie may legitimately go the other way; compare up to the first one
        upto = int(close[0]) if close.size else 330
        assert upto > 315, upto  # both bucket boundaries (256 keys at step 55, 512 at step 311) are inside the compared range
        assert a0[:upto] == ref[:upto], [i for i, (x, y) in enumerate(zip(a0, ref)) if x != y][:3]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_rows_flash_attention_on_a_long_left_padded_prompt(golden, dev, dtype):
    """Prefill of a 205-row prompt with 5 left-padding rows through the causal flash kernel (two 128-row workgroups, four
    64-key tiles, `valid_from` = 5): last-row logits against the oracle's masked prefill."""
    from voice_tts_amd.gpt_engine import GptEngine

    g = golden("gpt_tiny.npz")
    cfg, W, orc = _tiny(g)
    gen = torch.Generator().manual_seed(5)
    prompt = torch.cat([torch.zeros(5, cfg["model_dim"]), torch.randn(200, cfg["model_dim"], generator=gen) * 0.5])
    mask = torch.cat([torch.zeros(5, dtype=torch.long), torch.ones(201, dtype=torch.long)])
    ref, _ = orc.prefill(prompt, mask, 8192)
    ref = ref.numpy()
    eng = GptEngine(cfg, dtype=dtype, max_seq=512, max_batch=1, device=dev).load_state_dict(W)
    eng.prefill(0, prompt, 5)
    got = eng.read_logits(0)
    tol = 3e-4 if dtype == "f32" else 5e-2
    assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()
    if dtype == "f32":
        assert int(got.argmax()) == int(ref.argmax())
