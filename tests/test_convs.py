"""voice-tts_amd/convs.py (the glue stages' convolutions as GEMM forms) against `F.conv1d` / `F.conv2d` on every
configuration the glue uses: 1x1, k = 3 'same', strided, dilated, depthwise k = 15 / 7, the resampler's strided filter bank,
the conformer's stride-2 Conv2d, CAM++'s (stride, 1) Conv2d with padding."""
import pytest
import torch
import torch.nn.functional as F

from voice_tts_amd.convs import conv1d, conv2d


@pytest.mark.parametrize("B,Cin,Cout,T,k,stride,padding,dilation,groups", [
    (2, 16, 24, 37, 1, 1, 0, 1, 1),     # pointwise / 1x1
    (1, 12, 12, 50, 3, 1, 1, 1, 1),     # length regulator
    (1, 8, 20, 41, 5, 1, 0, 2, 1),      # WaveNet in_layers (dilated, caller pads)
    (1, 10, 6, 33, 5, 2, 2, 1, 1),      # CAM++ tdnn (stride 2, padding 2)
    (1, 6, 9, 40, 3, 1, 2, 2, 1),       # CAM++ cam_layer.linear_local (dilation 2)
    (1, 14, 14, 29, 15, 1, 7, 1, 14),   # conformer depthwise k = 15
    (2, 14, 14, 29, 7, 1, 3, 1, 14),    # codec dwconv k = 7
    (3, 1, 11, 64, 9, 4, 0, 1, 1),      # sinc resampler: one input channel, a bank of `new` filters, stride `orig`
    (1, 5, 7, 1, 1, 1, 0, 1, 1),        # T = 1
])
def test_conv1d_matches_torch(B, Cin, Cout, T, k, stride, padding, dilation, groups):
    g = torch.Generator().manual_seed(T * 131 + k)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin // groups, k, generator=g)
    b = torch.randn(Cout, generator=g)
    ref = F.conv1d(x, w, b, stride, padding, dilation, groups)
    got = conv1d(x, w, b, stride, padding, dilation, groups)
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    assert torch.allclose(conv1d(x, w, None, stride, padding, dilation, groups), F.conv1d(x, w, None, stride, padding, dilation, groups), rtol=1e-5,
                          atol=1e-5 * float(ref.abs().max()))
    # a non-contiguous input (the callers hand in transposed views)
    xt = x.transpose(1, 2).contiguous().transpose(1, 2)
    assert torch.allclose(conv1d(xt, w, b, stride, padding, dilation, groups), ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("B,Cin,Cout,H,W,kh,kw,stride,padding", [
    (1, 1, 8, 21, 30, 3, 3, 2, 0),          # conformer Conv2dSubsampling2
    (2, 4, 6, 16, 19, 3, 3, (2, 1), 1),     # CAM++ FCM block, strided over frequency
    (1, 3, 5, 10, 12, 3, 3, 1, 1),          # CAM++ FCM block
    (1, 4, 7, 9, 11, 1, 1, (2, 1), 0),      # CAM++ shortcut
])
def test_conv2d_matches_torch(B, Cin, Cout, H, W, kh, kw, stride, padding):
    g = torch.Generator().manual_seed(H * 17 + W)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, kh, kw, generator=g)
    b = torch.randn(Cout, generator=g)
    for bias in (b, None):
        ref = F.conv2d(x, w, bias, stride, padding)
        got = conv2d(x, w, bias, stride, padding)
        assert got.shape == ref.shape
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
