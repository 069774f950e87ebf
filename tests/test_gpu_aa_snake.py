"""GPU parity: HIP anti-aliased SnakeBeta (through the C ABI) vs oracle and reference fixtures."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


TAGS = ["c3_t1", "c3_t2", "c5_t5", "c4_t11", "c24_t64", "c6_t300", "c2_t4097"]


@pytest.mark.parametrize("tag", TAGS)
def test_vs_reference_fixture(golden, dev, tag):
    from voice_tts_amd import aa_activation as AA

    g = golden("aa_snake.npz")
    x = torch.from_numpy(g[f"x_{tag}"]).to(dev)
    y = AA.forward(x, torch.from_numpy(g["up_filter"]), torch.from_numpy(g["down_filter"]),
                   torch.from_numpy(g[f"la_{tag}"]), torch.from_numpy(g[f"lb_{tag}"]))
    ref = torch.from_numpy(g[f"y_{tag}"])
    assert y.shape == ref.shape and y.dtype == torch.float32 and y.is_cuda
    err = (y.cpu() - ref).abs().max().item()
    assert err <= 4e-6 * max(1.0, ref.abs().max().item()), err  # fp32: tolerance 4e-6 relative to max|y|


@pytest.mark.parametrize("B,C,T", [(1, 768, 400), (2, 24, 25600), (1, 7, 1023), (1, 5, 1024), (1, 5, 1025), (3, 2, 2049)])
def test_vs_oracle_seeded(dev, B, C, T):
    from oracle import vocoder as OV
    from voice_tts_amd import aa_activation as AA

    g = torch.Generator().manual_seed(B * 1000 + C * 10 + T)
    x = torch.randn(B, C, T, generator=g) * 2.0
    la = torch.randn(C, generator=g) * 0.5
    lb = torch.randn(C, generator=g) * 0.5
    f = torch.from_numpy(OV.kaiser_sinc_filter12())
    y = AA.forward(x.to(dev), f, f, la, lb).cpu()
    ref = OV.aa_snake(x, la, lb)
    assert (y - ref).abs().max().item() <= 4e-6 * max(1.0, ref.abs().max().item())


def test_empty_and_errors(dev):
    from voice_tts_amd import aa_activation as AA

    f = AA.kaiser_sinc_filter12()
    y = AA.forward(torch.zeros(1, 3, 0, device=dev), f, f, torch.zeros(3), torch.zeros(3))
    assert y.shape == (1, 3, 0)
    with pytest.raises(RuntimeError):
        AA.forward(torch.zeros(1, 3, 8, device=dev, dtype=torch.float64), f, f, torch.zeros(3), torch.zeros(3))
    with pytest.raises(RuntimeError):
        AA.forward(torch.zeros(1, 3, 8), f, f, torch.zeros(3), torch.zeros(3))


def test_module_mirror_matches_reference_interface(dev):
    """`Activation1d(SnakeBeta-like)` drop-in: same call shape as activation1d.py:35-77."""
    from oracle import vocoder as OV
    from voice_tts_amd import aa_activation as AA

    class SnakeBetaParams(torch.nn.Module):
        def __init__(self, C):
            super().__init__()
            self.alpha = torch.nn.Parameter(torch.linspace(-0.5, 0.5, C))
            self.beta = torch.nn.Parameter(torch.linspace(0.3, -0.3, C))
            self.alpha_logscale = True

    act = AA.Activation1d(SnakeBetaParams(6)).to(dev)
    x = torch.randn(2, 6, 333, generator=torch.Generator().manual_seed(3))
    y = act(x.to(dev)).cpu()
    ref = OV.aa_snake(x, act.act.alpha.detach().cpu(), act.act.beta.detach().cpu())
    assert (y - ref).abs().max().item() <= 4e-6 * max(1.0, ref.abs().max().item())
