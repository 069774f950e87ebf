"""Row N2 on the device: the conditioning encoders (conformer + perceiver, with their convolutions as GEMM forms -- convs.py) against
the fixture of the reference's OWN ConformerEncoder / PerceiverResampler (tests/golden/conditioning_tiny.npz), not against the same
code on the CPU: full-length and ragged batches, get_conditioning, get_emovec / merge_emovec, and the per-request entry
`encode_prompt` with and without a separate emotion prompt."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case(golden):
    import voice_tts_amd.conditioning as CD

    g = golden("conditioning_tiny.npz")
    cfg = CD.tiny_cond_cfg()
    return g, CD.Conditioning(CD.make_cond_weights(cfg, seed=int(g["seed"])), cfg, "cuda:0")


def _close(got, want, tol=5e-5):
    want = torch.from_numpy(np.asarray(want))
    got = got.cpu()
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err <= tol * max(1.0, want.abs().max().item()), err


def test_device_conditioning_vs_reference_fixture(case):
    g, m = case
    dev = m.device
    spk, emo = torch.from_numpy(g["spk"]).to(dev), torch.from_numpy(g["emo"]).to(dev)
    T = spk.shape[1]
    full = lambda x: torch.tensor([x.shape[1]] * x.shape[0], device=dev)
    y, mask = m.conformer(spk, full(spk), "conditioning_encoder.", m.cfg["condition_module"])
    _close(y, g["enc_full"])
    lens = torch.from_numpy(g["lens_ragged"]).to(dev)
    y, mask = m.conformer(spk, lens, "conditioning_encoder.", m.cfg["condition_module"])
    assert np.array_equal(mask.cpu().numpy(), g["mask_ragged"])
    keep = torch.from_numpy(g["mask_ragged"]).squeeze(1).unsqueeze(-1)
    _close(y.cpu() * keep, torch.from_numpy(g["enc_ragged"]) * keep)
    _close(m.get_conditioning(spk.transpose(1, 2), full(spk)), g["cond_full"])
    _close(m.get_conditioning(spk.transpose(1, 2), lens), g["cond_ragged"])
    _close(m.get_emovec(spk, full(spk)), g["emovec_spk"])
    _close(m.get_emovec(emo, full(emo)), g["emovec_emo"])
    _close(m.merge_emovec(spk, emo, full(spk), full(emo), alpha=0.7), g["merged_alpha07"])
    # the per-request entry (infer_v2.py:629-635) passes `shape[-1]` of the [1, T, F] features as the length, as the reference does (1024 in
    # production: no frame is masked; 21 < T = 23 in this twin: the last frames ARE masked) -- it must equal the calls above at that length
    F_ = spk.shape[-1]
    lf = torch.tensor([F_], device=dev)
    cond32, emovec = m.encode_prompt(spk[:1], emo[:1], 0.7)
    assert torch.equal(cond32, m.get_conditioning(spk[:1].transpose(1, 2), lf)[0])
    assert torch.equal(emovec, m.merge_emovec(spk[:1], emo[:1], lf, lf, alpha=0.7))
    cond32, emovec = m.encode_prompt(spk[:1], None, 1.0)
    assert torch.equal(emovec, m.get_emovec(spk[:1], lf))  # merge_emovec(spk, spk, 1.0) = the speaker's own vector, one encoder pass
