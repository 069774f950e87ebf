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
