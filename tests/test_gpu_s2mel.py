"""Row N1 on the device: the torch-glue s2mel stage on cuda:0 vs the reference fixture (same injected CFM noise)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_s2mel_on_gpu_vs_reference_fixture(golden):
    import voice_tts_amd.s2mel as S2

    g = golden("s2mel_tiny.npz")
    dev = torch.device("cuda:0")
    cfg = S2.tiny_s2mel_cfg(gpt_dim=1280, semantic_dim=1024, lr_in_channels=1024, codebook_size=8194)
    m = S2.S2Mel(S2.make_s2mel_weights(cfg, seed=int(g["seed"])), cfg, device=dev)
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    n = g["codes"].shape[1]
    mel = m(t("latent"), t("codes"), torch.tensor([n], device=dev), t("prompt_condition"), t("ref_mel"), t("style"),
            n_timesteps=int(g["n_steps"]), inference_cfg_rate=0.7, noise=t("noise")).cpu()
    ref = torch.from_numpy(g["mel"])
    assert mel.shape == ref.shape
    assert (mel - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,H,T", [(2, 8, 2322), (1, 3, 301), (1, 1, 64), (2, 2, 65), (1, 2, 1)])
def test_attn_full_f32_vs_torch_sdpa(B, H, T):
    """HIP fp32-MFMA flash attention of the DiT vs torch's fp32 SDPA (the reference op, gpt_fast/model.py:303)."""
    from voice_tts_amd.s2mel import attn_full

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 100 + T)
    q, k, v = (torch.randn(B, T, H, 64, generator=g).to(dev) for _ in range(3))
    q = q * 2.0  # sharper softmax
    out = attn_full(q, k, v)
    ref = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2).double(), k.transpose(1, 2).double(), v.transpose(1, 2).double()).transpose(1, 2).float()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # strided views (as the DiT hands them over: slices of the fused wqkv output)
    qkv = torch.randn(B, T, 3 * H * 64, generator=g).to(dev)
    q2, k2, v2 = (t.reshape(B, T, H, 64) for t in qkv.split(H * 64, dim=-1))
    out2 = attn_full(q2, k2, v2)
    ref2 = torch.nn.functional.scaled_dot_product_attention(q2.transpose(1, 2), k2.transpose(1, 2), v2.transpose(1, 2)).transpose(1, 2)
    assert (out2 - ref2).abs().max().item() <= 2e-5 * max(1.0, ref2.abs().max().item())
