"""Row N3 on the device: continuous batching over the decode slots gives every sequence the tokens it gets alone."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batched_decode_equals_one_at_a_time(golden):
    import voice_tts_amd.weights as WR
    from voice_tts_amd.pipeline import HotPath

    dev = torch.device("cuda:0")
    gcfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    Wg = WR.make_gpt_weights(gcfg, seed=7)
    Wb = WR.make_bigvgan_weights(WR.tiny_bigvgan_cfg(64), seed=8)
    g = torch.Generator().manual_seed(31)
    segs = []
    for i, (rows, pad, n) in enumerate([(20, 0, 30), (33, 2, 12), (9, 0, 41), (27, 5, 25), (14, 0, 8), (40, 0, 33), (11, 1, 19)]):
        e = torch.randn(rows, 128, generator=g) * 0.5
        e[:pad] = 0
        segs.append((e.to(dev), pad, n))

    def run(max_batch, fixed):
        hp = HotPath(gpt_cfg=gcfg, bigvgan_cfg=WR.tiny_bigvgan_cfg(64), dtype="f32", device=dev, max_batch=max_batch, max_seq=128, max_frames=32).load(Wg, Wb)
        ids = hp.generate_many(segs, fixed_length=fixed, sync_every=8)
        return [np.asarray(x) for x in ids], hp.last_sched_stats

    for fixed in (True, False):
        alone, _ = run(1, fixed)
        for mb in (3, 4):
            got, stats = run(mb, fixed)
            assert len(got) == len(alone)
            for a, b in zip(alone, got):
                assert a.tolist() == b.tolist()
            assert stats["refills"] >= len(segs) - mb
        if fixed:
            assert [len(x) for x in alone] == [n for _, _, n in segs]
