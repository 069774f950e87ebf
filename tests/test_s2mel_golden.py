"""Row N1 (PyTorch glue): the s2mel stage restated in voice-tts_amd/s2mel.py against outputs of the reference's own
MyModel / InterpolateRegulator / CFM(DiT + WaveNet) / FactorizedVectorQuantize (tests/golden/s2mel_tiny.npz).
Pure torch, so it runs on the CPU here and on the GPU in test_gpu_s2mel."""
import numpy as np
import pytest
import torch

import voice_tts_amd.s2mel as S2


FIXTURES = {"s2mel_tiny.npz": {}, "s2mel_hd64.npz": dict(hidden_dim=128, num_heads=2, wavenet_hidden=128, depth=3)}


def _model(g, device="cpu", **kw):
    cfg = S2.tiny_s2mel_cfg(gpt_dim=1280, semantic_dim=1024, lr_in_channels=1024, codebook_size=8194, **kw)
    W = S2.make_s2mel_weights(cfg, seed=int(g["seed"]))
    return S2.S2Mel(W, cfg, device=device), cfg


def _t(g, k, device="cpu"):
    return torch.from_numpy(g[k]).to(device)


@pytest.mark.parametrize("fx", list(FIXTURES))
def test_s2mel_stages_vs_reference(golden, fx):
    g = golden(fx)
    m, cfg = _model(g, **FIXTURES[fx])
    lat = m.gpt_layer(_t(g, "latent"))
    assert torch.allclose(lat, _t(g, "gpt_layer_out"), atol=2e-5)
    emb = m.vq2emb(_t(g, "codes"))
    assert torch.allclose(emb, _t(g, "vq_emb"), atol=1e-5)
    n = g["codes"].shape[1]
    cond = m.length_regulator(emb + lat, torch.tensor([int(n * 1.72)]))
    assert cond.shape == _t(g, "cond").shape and torch.allclose(cond, _t(g, "cond"), atol=2e-5)
    # one DiT evaluation (transformer with RoPE + U-ViT skips + AdaLN-RMSNorm, WaveNet head, final layer)
    cat = torch.cat([_t(g, "prompt_condition"), cond], dim=1)
    T = cat.shape[1]
    one = m.dit(_t(g, "noise"), torch.zeros(1, 80, T), torch.tensor([T]), torch.tensor([0.3]), _t(g, "style"), cat)
    ref = _t(g, "dit_one")
    assert (one - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("fx", list(FIXTURES))
def test_s2mel_end_to_end_vs_reference(golden, fx):
    g = golden(fx)
    m, cfg = _model(g, **FIXTURES[fx])
    n = g["codes"].shape[1]
    mel = m(_t(g, "latent"), _t(g, "codes"), torch.tensor([n]), _t(g, "prompt_condition"), _t(g, "ref_mel"), _t(g, "style"),
            n_timesteps=int(g["n_steps"]), inference_cfg_rate=0.7, noise=_t(g, "noise"))
    ref = _t(g, "mel")
    assert mel.shape == ref.shape == (1, 80, int(n * 1.72))
    assert (mel - ref).abs().max().item() <= 5e-5 * max(1.0, ref.abs().max().item())


def test_weight_norm_checkpoint_spelling_loads(golden):
    """A checkpoint that still carries weight_g / weight_v (as s2mel.pth does) folds to the same tensors."""
    g = golden("s2mel_tiny.npz")
    m, cfg = _model(g)
    W = S2.make_s2mel_weights(cfg, seed=int(g["seed"]))
    sd = {}
    for k, v in W.items():
        if k.endswith("conv.conv.weight") or k.endswith("final_layer.linear.weight"):
            sd[k[:-6] + "weight_g"] = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
            sd[k[:-6] + "weight_v"] = 3.0 * v
        else:
            sd[k] = v
    m2 = S2.S2Mel(sd, cfg)
    for k in W:
        assert torch.allclose(m2.W[k], m.W[k], atol=1e-6), k


def test_wavenet_gemm_form_equals_the_conv_form():
    """The pipeline's WaveNet (accumulating GEMMs, residual biases carried into the next conv's bias, skip biases added once)
    is the reference form re-associated: same values up to fp32 rounding."""
    import voice_tts_amd.s2mel as S2

    cfg = S2.tiny_s2mel_cfg()
    m = S2.S2Mel(S2.make_s2mel_weights(cfg, seed=13), cfg, "cpu")
    g = torch.Generator().manual_seed(14)
    x = torch.randn(2, cfg["wavenet_hidden"], 37, generator=g)
    t2 = torch.randn(2, cfg["wavenet_hidden"], generator=g)
    want = m._wavenet(x.clone(), torch.ones(2, 1, 37), t2, True)
    got = m._wavenet_gemm(x.clone(), t2) + m.wn_out_bias[None, :, None]
    assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    # sequences as rows, one GEMM per tap over all sequences (what the GPU runs for dilation 1)
    rows = m._wavenet_rows(x.transpose(1, 2).contiguous(), t2).transpose(1, 2) + m.wn_out_bias[None, :, None]
    assert rows.shape == want.shape
    assert (rows - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())


def test_dit_head_on_the_needed_frames_only():
    """dit_step(out_from=k) -- the WaveNet head evaluated on frames [k - receptive field, T) -- equals the full evaluation on
    frames [k, T): what the CFM solver uses, since it zeroes the prompt frames after every step."""
    import voice_tts_amd.s2mel as S2

    cfg = S2.tiny_s2mel_cfg()
    m = S2.S2Mel(S2.make_s2mel_weights(cfg, seed=21), cfg, "cpu")
    g = torch.Generator().manual_seed(22)
    B, T, k = 2, 61, 27
    x = torch.randn(1, 80, T, generator=g)
    prompt_x = torch.randn(B, 80, T, generator=g)
    cond = torch.randn(B, T, cfg["content_dim"], generator=g)
    style = torch.randn(B, cfg["style_dim"], generator=g)
    ctx = m.dit_prepare(prompt_x, torch.tensor([T, T]), style, cond)
    t = torch.tensor([0.3, 0.3])
    full = m.dit_step(ctx, x, t)
    part = m.dit_step(ctx, x, t, out_from=k)
    assert part.shape == (B, 80, T - k)
    assert (part - full[:, :, k:]).abs().max().item() <= 1e-5 * max(1.0, full.abs().max().item())
