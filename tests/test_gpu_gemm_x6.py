"""csrc/gemm_x6.hip (row N1: the DiT / WaveNet linear layers as six exact bf16 MFMA partial products per fp32 product) against
fp64: every tile shape, ragged M and N, bias, accumulate (addmm_), strided rows (the WaveNet's row-shifted taps), and an error no
larger than the library's fp32 GEMM on the production shapes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,tile", [(300, 200, 64, 3), (300, 200, 64, 2), (700, 520, 192, 3), (4644, 1536, 512, 0), (4644, 512, 1536, 0),
                                        (257, 80, 512, 0), (1, 1024, 512, 0), (2322, 3072, 512, 2)])
def test_gemm_x6_matches_fp64(M, N, K, tile):
    from voice_tts_amd import gemm as G

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))  # rows of different scales
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = (x.double() @ w.double().t() + b.double())
    pl = G.PackedLinear(w.to(dev), b.to(dev))
    out = G.linear(x.to(dev), pl, tile=tile).cpu().double()
    lib = torch.nn.functional.linear(x.to(dev), w.to(dev), b.to(dev)).cpu().double()
    scale = ref.abs().max()
    e_x6, e_lib = float((out - ref).abs().max() / scale), float((lib - ref).abs().max() / scale)
    rms_x6, rms_lib = float(((out - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()), float(((lib - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    print(f"gemm_x6 {M}x{N}x{K} tile {tile}: max err {e_x6:.2e} (library fp32 {e_lib:.2e}), rms rel {rms_x6:.2e} (library {rms_lib:.2e})")
    assert e_x6 <= 2e-6 and rms_x6 <= max(2e-7, 2.0 * rms_lib)


def test_gemm_x6_accumulate_strided_rows_and_no_bias():
    from voice_tts_amd import gemm as G

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    T, K, N = 900, 128, 256
    buf = torch.randn(T + 4, K, generator=g).to(dev)      # a reflect-padded row buffer: tap j = rows j .. j + T
    taps = [torch.randn(N, K, generator=g) / K ** 0.5 for _ in range(5)]
    pls = [G.PackedLinear(t.to(dev)) for t in taps]
    out = torch.randn(T, N, generator=g).to(dev)
    ref = out.cpu().double()
    planes = G.split(buf)  # ONE split of the padded buffer; tap j = rows [j, j + T) of it
    for j, t in enumerate(taps):
        G.linear(planes, pls[j], out=out, accumulate=True, row0=j, rows=T)
        ref = ref + buf[j:j + T].cpu().double() @ t.double().t()
    assert float((out.cpu().double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    # a column-sliced view as input (row stride > K) and a 3-D input
    wide = torch.randn(500, 2 * K, generator=g).to(dev)
    y = G.linear(wide[:, K:], pls[0])
    assert torch.allclose(y.cpu().double(), wide[:, K:].cpu().double() @ taps[0].double().t(), atol=2e-6 * float(y.abs().max()))
    x3 = torch.randn(2, 333, K, generator=g).to(dev)
    y3 = G.linear(x3, pls[1])
    assert y3.shape == (2, 333, N)
    assert torch.allclose(y3.cpu().double(), x3.cpu().double() @ taps[1].double().t(), atol=2e-6 * float(y3.abs().max()))
