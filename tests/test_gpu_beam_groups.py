"""Beam groups (the served default for several text segments at once, infer_v2.py:598-606,616): a group's tokens, source
beams and scores are the ones it gets when it decodes alone -- bit for bit -- whatever groups step beside it, whichever group
it lands in, with free-running draws (per-segment random streams) and with forced draws."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NB = 3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def tiny(dev):
    import voice_tts_amd.weights as WR
    from voice_tts_amd.gpt_engine import GptEngine

    cfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    W = WR.make_gpt_weights(cfg, seed=5, head_scale=50.0)
    W["mel_head.bias"] = W["mel_head.bias"].clone()
    W["mel_head.bias"][8193] += 24.0  # eos reachable: some groups collect hypotheses and finish early
    eng = GptEngine(cfg, dtype="bf16", max_seq=160, max_batch=15, device=dev).load_state_dict(W)  # wide engine: 5 groups of 3
    g = torch.Generator().manual_seed(77)
    prompts = []
    for rows, pad in ((20, 0), (33, 2), (9, 0), (27, 5), (41, 0), (14, 1), (25, 0)):
        e = torch.randn(rows, 128, generator=g) * 0.5
        e[:pad] = 0
        prompts.append((e.to(dev), pad))
    return cfg, W, eng, prompts


def _snap(eng, max_new, group):
    ids, done, score, bs, lt, src = eng.beam_read(max_new, group=group)
    return ids.tolist(), done, score, bs.tolist(), lt.tolist(), src.tolist()


def _alone(eng, prompt, stream, n, chunk, **kw):
    """One group at a time (group 0), read after every `chunk` steps."""
    eng.prefill(0, *prompt)
    eng.beam_begin(NB, group=0, rng_stream=stream)
    out = []
    for _ in range(0, n, chunk):
        eng.beam_decode(chunk, groups=1, seed=11, **kw)
        out.append(_snap(eng, n, 0))
    return out


def test_groups_together_equal_groups_alone_free_running(tiny):
    cfg, W, eng, prompts = tiny
    n, chunk = 48, 8
    alone = [_alone(eng, prompts[i], i, n, chunk) for i in range(5)]
    print("done flags per chunk:", [[int(c[1]) for c in a] for a in alone], "final lengths:", [len(a[-1][0]) for a in alone])
    assert any(a[-1][1] for a in alone), "want finished groups (scorer done) in this test"
    assert len({tuple(a[-1][0]) for a in alone}) == 5
    for order in ([0, 1, 2, 3, 4], [3, 0, 4, 2, 1]):  # segment order[g] lands in group g
        for g, i in enumerate(order):
            eng.prefill(g * NB, *prompts[i])
            eng.beam_begin(NB, group=g, rng_stream=i)
        for c in range(n // chunk):
            eng.beam_decode(chunk, groups=5, seed=11)
            for g, i in enumerate(order):
                assert _snap(eng, n, g) == alone[i][c], (order, g, i, c)


def test_groups_together_equal_groups_alone_forced_draws(tiny):
    """Forced draws (`ixtts_gpt_beam_force_group`): per step and group 2*NB flat picks from a seeded generator over that
    group's beams and the 40 most likely tokens of the vocabulary's head (picks outside a beam's processed support score
    -inf on both sides alike)."""
    cfg, W, eng, prompts = tiny
    V, steps = cfg["number_mel_codes"], 24
    rng = np.random.default_rng(3)
    picks = rng.integers(0, 40, size=(3, steps, 2 * NB)) + V * rng.integers(0, NB, size=(3, steps, 2 * NB))
    picks[:, 0, :] %= V  # first step: only beam 0 is live
    for s in range(steps):  # distinct flat picks within a step, as sampling without replacement gives
        for g in range(3):
            while len(set(picks[g, s].tolist())) < 2 * NB:
                picks[g, s] = rng.integers(0, 40, size=2 * NB) + V * (rng.integers(0, NB, size=2 * NB) if s else 0)
    alone = []
    for i in range(3):
        eng.prefill(0, *prompts[i])
        eng.beam_begin(NB, group=0)
        tr = []
        for s in range(steps):
            eng.beam_force(picks[i, s], group=0)
            eng.beam_decode(1, groups=1, suppress_stop=True)
            tr.append(_snap(eng, steps, 0))
        alone.append(tr)
    for g in range(3):
        eng.prefill(g * NB, *prompts[g])
        eng.beam_begin(NB, group=g)
    for s in range(steps):
        for g in range(3):
            eng.beam_force(picks[g, s], group=g)
        eng.beam_decode(1, groups=3, suppress_stop=True)
        for g in range(3):
            assert _snap(eng, steps, g) == alone[g][s], (g, s)


def test_scheduler_refills_and_parks_groups_same_tokens(tiny, dev):
    """Seven segments of different lengths through three groups (continuous batching: a finished group is refilled while
    the others are mid-sequence, parked when the queue is empty) == the same segments one at a time."""
    import voice_tts_amd.weights as WR
    from voice_tts_amd.pipeline import HotPath

    cfg, W, eng, prompts = tiny
    lens = [30, 12, 41, 25, 8, 33, 19]
    segs = [(e, p, n) for (e, p), n in zip(prompts, lens)]
    Wb = WR.make_bigvgan_weights(WR.tiny_bigvgan_cfg(64), seed=8)

    def run(max_batch):
        hp = HotPath(gpt_cfg=cfg, bigvgan_cfg=WR.tiny_bigvgan_cfg(64), dtype="bf16", device=dev, max_batch=max_batch, max_seq=160, max_frames=32).load(W, Wb)
        out = hp.generate_beams_many(segs, num_beams=NB, sync_every=8, seed=5)
        return [np.asarray(x).tolist() for x in out], hp.last_sched_stats

    five, st5 = run(15)  # five groups on the wide engine
    from voice_tts_amd.scheduler import BeamGroupScheduler, Segment

    # the same engine shape, one group at a time (the reference's order: segment after segment)
    hp = HotPath(gpt_cfg=cfg, bigvgan_cfg=WR.tiny_bigvgan_cfg(64), dtype="bf16", device=dev, max_batch=15, max_seq=160, max_frames=32).load(W, Wb)
    turn = [None] * len(segs)
    BeamGroupScheduler(hp.gpt, NB, max_groups=1, sync_every=8).run([Segment(0, i, e, p, n) for i, (e, p, n) in enumerate(segs)],
                                                                   lambda seg, ids, sc: turn.__setitem__(seg.index, np.asarray(ids).tolist()), seed=5)
    assert five == turn
    assert st5["refills"] == 2 and all(len(x) <= n + 1 for x, n in zip(five, lens))
    three, st3 = run(9)  # three groups: more refills, parked groups at the end
    assert three == turn and st3["refills"] == 4


def test_group_api_argument_errors(tiny, dev):
    import voice_tts_amd.weights as WR
    from voice_tts_amd._lib import IxttsError
    from voice_tts_amd.gpt_engine import GptEngine

    cfg, W, eng, prompts = tiny
    with pytest.raises(IxttsError):
        eng.beam_begin(NB, group=5)  # 6 groups of 3 do not fit 15 slots
    small = GptEngine(cfg, dtype="f32", max_seq=96, max_batch=3, device=dev).load_state_dict(W)
    small.prefill(0, *prompts[0])
    small.beam_begin(NB)
    with pytest.raises(IxttsError):
        small.beam_decode(1, groups=2)  # several groups need a wide engine
    small.beam_decode(2)
    assert len(small.beam_read(8)[0]) >= 2
