"""Wide engines (max_batch 5..16): the decode GEMVs on the bf16 matrix cores (csrc/gpt_wide.h).  A sequence's arithmetic must not
depend on its company (bit-identical logits and tokens alone or among 8), must agree with the register GEMVs of the narrow
engine to rounding, and must stay within the stated bf16 bounds of the fp32 CPU oracle at the benchmarked shape."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _tiny():
    import voice_tts_amd.weights as WR

    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    return cfg, WR.make_gpt_weights(cfg, seed=7, head_scale=50.0)


def _prompts(n, D, seed=31):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        rows, pad = 9 + 5 * i, (i % 3)
        e = torch.randn(rows, D, generator=g) * 0.5
        e[:pad] = 0
        out.append((e, pad))
    return out


def test_wide_tokens_and_logits_do_not_depend_on_company(dev):
    from voice_tts_amd.gpt_engine import GptEngine

    cfg, W = _tiny()
    P = _prompts(8, 128)
    eng = GptEngine(cfg, dtype="bf16", max_seq=160, max_batch=8, device=dev).load_state_dict(W)
    n = 40
    alone = []
    for e, pad in P:
        eng.prefill(0, e, pad)
        first = eng.read_logits(0).copy()
        eng.decode(1, n, repetition_penalty=10.0, suppress_stop=True)
        alone.append((first, eng.read(0)[0][:n].tolist(), eng.read_logits(0).copy()))
    for B in (8, 5):
        for b in range(B):
            eng.prefill(b, *P[b])
        firsts = [eng.read_logits(b).copy() for b in range(B)]
        eng.decode(B, n, repetition_penalty=10.0, suppress_stop=True)
        for b in range(B):
            assert np.array_equal(firsts[b], alone[b][0]), (B, b)
            assert eng.read(b)[0][:n].tolist() == alone[b][1], (B, b)
            assert np.array_equal(eng.read_logits(b), alone[b][2]), (B, b)  # bit for bit after 40 steps
    # in another slot order too (slot index is not part of the arithmetic)
    for b in range(8):
        eng.prefill(b, *P[7 - b])
    eng.decode(8, n, repetition_penalty=10.0, suppress_stop=True)
    for b in range(8):
        assert eng.read(b)[0][:n].tolist() == alone[7 - b][1]
    # all 16 columns of the MFMA in use (the largest engine): each prompt twice, every copy as alone on the 8-slot engine
    big = GptEngine(cfg, dtype="bf16", max_seq=160, max_batch=16, device=dev).load_state_dict(W)
    for b in range(16):
        big.prefill(b, *P[b % 8])
    big.decode(16, n, repetition_penalty=10.0, suppress_stop=True)
    for b in range(16):
        assert big.read(b)[0][:n].tolist() == alone[b % 8][1], b
        assert np.array_equal(big.read_logits(b), alone[b % 8][2]), b


def test_wide_agrees_with_the_register_gemvs_and_the_bf16_rounded_oracle(dev):
    """Same weights through both engines: logits agree to fp32 summation-order noise for the LN / attention inputs (hi + lo split)
    plus the one bf16 rounding of the ff activations; greedy tokens equal; and the wide engine passes the bf16 tiny-twin check of
    test_gpu_gpt.py (oracle on bf16-rounded weights)."""
    from oracle import gpt as OG
    from voice_tts_amd.gpt_engine import GptEngine

    cfg, W = _tiny()
    P = _prompts(3, 128, seed=5)
    narrow = GptEngine(cfg, dtype="bf16", max_seq=160, max_batch=3, device=dev).load_state_dict(W)
    wide = GptEngine(cfg, dtype="bf16", max_seq=160, max_batch=16, device=dev).load_state_dict(W)
    n = 32
    outs = []
    for eng in (narrow, wide):
        for b, (e, pad) in enumerate(P):
            eng.prefill(b, e, pad)
        l0 = [eng.read_logits(b).copy() for b in range(3)]
        eng.decode(3, n, repetition_penalty=10.0, suppress_stop=True)
        outs.append((l0, [eng.read(b)[0][:n].tolist() for b in range(3)], [eng.read_logits(b).copy() for b in range(3)]))
    scale = max(float(np.abs(x).max()) for x in outs[0][0])
    for b in range(3):
        assert np.abs(outs[0][0][b] - outs[1][0][b]).max() <= 1e-5 * scale  # prefill rows + head: hi + lo is fp32-faithful
        assert outs[0][1][b] == outs[1][1][b]
        assert np.abs(outs[0][2][b] - outs[1][2][b]).max() <= 4e-3 * scale  # ff travels as bf16 in the wide engine
    mats = ("c_attn.weight", "c_proj.weight", "c_fc.weight", "mel_head.weight")
    Wq = {k: (v.to(torch.bfloat16).to(torch.float32) if k.endswith(mats) else v) for k, v in W.items()}
    orc = OG.GptOracle(Wq, cfg["layers"], cfg["heads"])
    e, pad = P[0]
    mask = torch.ones(e.shape[0] + 1, dtype=torch.long)
    mask[:pad] = 0
    ids, margins, logits = OG.generate_greedy(orc, e, mask, 24, return_logits=True, suppress_stop=True)
    assert np.abs(outs[1][0][0] - logits[0].numpy()).max() <= 1e-2 * np.abs(logits[0].numpy()).max()
    assert outs[1][1][0][:4] == ids[:4] and np.mean(np.array(outs[1][1][0][:24]) == np.array(ids)) >= 0.5


def test_scheduler_with_8_slots_equals_one_at_a_time(dev):
    """tests/test_gpu_scheduler.py at 8 slots: continuous batching over a wide engine gives every sequence the tokens it gets
    alone on that engine (refills, idle slots and draining included)."""
    import voice_tts_amd.weights as WR
    from voice_tts_amd.pipeline import HotPath

    cfg, W = _tiny()
    Wb = WR.make_bigvgan_weights(WR.tiny_bigvgan_cfg(64), seed=8)
    g = torch.Generator().manual_seed(77)
    segs = []
    for i in range(19):
        rows, pad, n = 8 + (7 * i) % 31, i % 4, 6 + (11 * i) % 37
        e = torch.randn(rows, 128, generator=g) * 0.5
        e[:pad] = 0
        segs.append((e.to(dev), pad, n))
    hp = HotPath(gpt_cfg=cfg, bigvgan_cfg=WR.tiny_bigvgan_cfg(64), dtype="bf16", device=dev, max_batch=8, max_seq=128, max_frames=32).load(W, Wb)
    for fixed in (True, False):
        alone = [np.asarray(hp.generate([(e, p)], n, fixed_length=fixed)[0]) for e, p, n in segs]
        if not fixed:
            stop = cfg["stop_mel_token"]
            alone = [a[: int(np.nonzero(a == stop)[0][0]) + 1] if (a == stop).any() else a for a in alone]
        got = hp.generate_many(segs, fixed_length=fixed, sync_every=8)
        assert hp.last_sched_stats["refills"] >= len(segs) - 8
        for a, b in zip(alone, got):
            assert a.tolist() == np.asarray(b).tolist()


def test_wide_engine_at_the_benchmarked_shape_vs_oracle(dev):
    """Full size (24 x 1280, bf16), 8 slots carrying the two benchmark prompts + 6 fillers, 1100 free-running steps: the same
    stated bounds as test_bench_shape_bf16_1100_steps_vs_oracle (logits <= 1.5e-2 of the scale at every read point, >= 95 %
    agreement with the fp32 oracle's greedy choice)."""
    import voice_tts_amd.weights as WR
    from oracle import gpt as OG
    from voice_tts_amd.gpt_engine import GptEngine

    W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
    orc = OG.GptOracle(W, WR.GPT_CFG["layers"], WR.GPT_CFG["heads"])
    g = torch.Generator().manual_seed(100)
    prompts = []
    for text in (torch.randint(2, 12000, (100,), generator=g), torch.cat((torch.tensor([0, 1, 0]), torch.randint(2, 12000, (77,), generator=g)))):
        conds = torch.randn(34, 1280, generator=g) * 0.5
        fake, embeds, mask = orc.prepare_gpt_inputs(conds, text)
        prompts.append((embeds, mask, int((mask == 0).sum())))
    N = 1100
    eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=137 + N + 64, max_batch=8, device=dev).load_state_dict(W)
    slots = [0, 5]  # the checked sequences sit among fillers
    for b in range(8):
        emb, mask, pad = prompts[slots.index(b)] if b in slots else prompts[b % 2]
        eng.prefill(b, emb[: emb.shape[0] - (0 if b in slots else 3 * b)] if pad == 0 else emb, pad)
    stops = sorted({1, 2, 64, 119, 120, 375, 376, 631, 887, N})
    got = {0: [eng.read_logits(b).copy() for b in slots]}
    done = 0
    for k in stops:
        eng.decode(8, k - done, repetition_penalty=10.0, suppress_stop=True)
        done = k
        got[k] = [eng.read_logits(b).copy() for b in slots]
    for j, b in enumerate(slots):
        ids = eng.read(b)[0][:N]
        emb, mask, pad = prompts[j]
        rows = OG.teacher_forced_logits(orc, emb, mask, ids.tolist())
        picks, margins = OG.greedy_choices(rows, len(mask), ids.tolist(), theta=10.0, suppress_stop=True)
        scale = float(rows.abs().max())
        err = max(np.abs(got[k][j] - rows[k].numpy()).max() for k in got) / scale
        agree = sum(int(picks[k] == int(ids[k])) for k in range(N)) / N
        print(f"wide bf16 slot {b}: logits rel err {err:.2e}, greedy agreement {agree:.4f}")
        assert err <= 1.5e-2 and agree >= 0.95, (b, err, agree)
