"""`F.conv1d` / `F.conv2d` of the PyTorch-hosted glue stages (conditioning encoders, s2mel length regulator, prompt-side
CAM++ / codec) as GEMM forms: strided tap views (`Tensor.unfold`, no copy until the GEMM) contracted against the reshaped
weight by `matmul` / `einsum` -- the same sums as the convolution, in the library GEMM's order.

Why not `F.conv*` on the device: on ROCm they go to MIOpen, which ships no gfx950 tuning database in this image and therefore
compiles its kernels the first time every (shape, stride, dilation, group) configuration is seen -- through comgr, into a
per-user cache -- and pages in a 0.9 GB library plus its databases to do so.  That is seconds to minutes on the FIRST request of
a fresh worker (profiles/r03_notes.md: "cold start") for convolutions that are a few MFLOP each.  The GEMM forms need nothing
beyond the BLAS library the linear layers already use.  On CPU tensors the same code runs (tests/test_convs.py holds both
forms to `F.conv*` on every configuration the glue uses)."""
import torch
import torch.nn.functional as F


def conv1d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    """x [B, Cin, T], w [Cout, Cin / groups, k] -> [B, Cout, T'] with `F.conv1d`'s semantics (zero padding).
    groups: 1, or depthwise (groups == Cin == Cout, one filter per channel)."""
    B, Cin, T = x.shape
    Cout, Cg, k = w.shape
    if k == 1 and stride == 1 and padding == 0 and groups == 1:
        y = torch.matmul(w[:, :, 0], x)  # [Cout, Cin] @ [B, Cin, T]
        return y if b is None else y + b.view(1, -1, 1)
    if padding:
        x = F.pad(x, (padding, padding))
    span = (k - 1) * dilation + 1
    taps = x.unfold(-1, span, stride)  # [B, Cin, T', span] view
    if dilation > 1:
        taps = taps[..., ::dilation]   # [B, Cin, T', k]
    if groups == 1:
        assert Cg == Cin
        # y[b, o, t] = sum_{c, j} w[o, c, j] * taps[b, c, t, j]: one GEMM [B*T', Cin*k] x [Cin*k, Cout]
        cols = taps.permute(0, 2, 1, 3).reshape(B, taps.shape[2], Cin * k)
        y = torch.matmul(cols, w.reshape(Cout, Cin * k).t()).transpose(1, 2)
    else:
        assert groups == Cin == Cout and Cg == 1, "grouped convolutions other than depthwise are not used by the glue"
        y = torch.einsum("bctk,ck->bct", taps, w[:, 0, :])
    return y if b is None else y + b.view(1, -1, 1)


def conv2d(x, w, b=None, stride=1, padding=0):
    """x [B, Cin, H, W], w [Cout, Cin, kh, kw] -> [B, Cout, H', W'] with `F.conv2d`'s semantics (zero padding, dilation 1,
    groups 1); stride / padding: int or (h, w)."""
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    B, Cin, H, Wd = x.shape
    Cout, Cg, kh, kw = w.shape
    assert Cg == Cin
    if ph or pw:
        x = F.pad(x, (pw, pw, ph, ph))
    taps = x.unfold(2, kh, sh).unfold(3, kw, sw)  # [B, Cin, H', W', kh, kw] view
    Ho, Wo = taps.shape[2], taps.shape[3]
    cols = taps.permute(0, 2, 3, 1, 4, 5).reshape(B, Ho * Wo, Cin * kh * kw)
    y = torch.matmul(cols, w.reshape(Cout, Cin * kh * kw).t())  # [B, H'*W', Cout]
    y = y.transpose(1, 2).reshape(B, Cout, Ho, Wo)
    return y if b is None else y + b.view(1, -1, 1, 1)
