"""Row N2: the conditioning-encoder glue against the reference's own ConformerEncoder / PerceiverResampler composed as
UnifiedVoice composes them (fixture: tests/golden/make_golden.py gen_conditioning).  fp32, tolerances relative to max|ref|."""
import numpy as np
import pytest
import torch

import voice_tts_amd.conditioning as CD


@pytest.fixture(scope="module")
def case(golden):
    g = golden("conditioning_tiny.npz")
    cfg = CD.tiny_cond_cfg()
    return g, CD.Conditioning(CD.make_cond_weights(cfg, seed=int(g["seed"])), cfg, "cpu")


def _close(got, want, tol=2e-5):
    want = torch.from_numpy(np.asarray(want))
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err <= tol * max(1.0, want.abs().max().item()), err


def test_conformer_full_and_ragged(case):
    g, m = case
    spk = torch.from_numpy(g["spk"])
    T = spk.shape[1]
    y, mask = m.conformer(spk, torch.tensor([T, T]), "conditioning_encoder.", m.cfg["condition_module"])
    _close(y, g["enc_full"])
    assert bool(mask.all()) and mask.shape[-1] == (T - 1) // 2
    y, mask = m.conformer(spk, torch.from_numpy(g["lens_ragged"]), "conditioning_encoder.", m.cfg["condition_module"])
    assert np.array_equal(mask.numpy(), g["mask_ragged"])
    keep = torch.from_numpy(g["mask_ragged"]).squeeze(1).unsqueeze(-1)  # padded frames carry no contract
    _close(y * keep, torch.from_numpy(g["enc_ragged"]) * keep)


def test_get_conditioning(case):
    g, m = case
    spk = torch.from_numpy(g["spk"]).transpose(1, 2)  # [B,1024,T] as inference_speech passes it (model_v2.py:684)
    T = spk.shape[-1]
    _close(m.get_conditioning(spk, torch.tensor([T, T])), g["cond_full"])
    _close(m.get_conditioning(spk, torch.from_numpy(g["lens_ragged"])), g["cond_ragged"])
    # the pipeline's length argument is spk_cond_emb.shape[-1] = 1024 (infer_v2.py:632,644): longer than T, i.e. no padding
    _close(m.get_conditioning(spk, torch.tensor([1024, 1024])), g["cond_full"])


def test_emovec_and_merge(case):
    g, m = case
    spk, emo = torch.from_numpy(g["spk"]), torch.from_numpy(g["emo"])
    full = lambda x: torch.tensor([x.shape[1]] * x.shape[0])
    _close(m.get_emovec(spk, full(spk)), g["emovec_spk"])
    _close(m.get_emovec(emo, full(emo)), g["emovec_emo"])
    _close(m.merge_emovec(spk, emo, full(spk), full(emo), alpha=0.7), g["merged_alpha07"])


def test_host_side_mask_decision_matches_the_device_one():
    """`lens_host` lets the encoders decide mask-free execution without reading the mask back (a hipGraph capture cannot): the
    predicate equals `bool(mask[:, :, 2::2].all())` for every (T, length)."""
    from voice_tts_amd.conditioning import Conditioning

    for T in range(1, 40):
        for n in range(0, 45):
            mask = (torch.arange(T).unsqueeze(0) < torch.tensor([n]).unsqueeze(1)).unsqueeze(1)[:, :, 2::2]
            assert bool(mask.all()) == Conditioning._kept_mask_is_full(T, [n]), (T, n)
