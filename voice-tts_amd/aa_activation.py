"""Seam 3: the reference's native activation ABI, backed by the HIP kernel.

Mirrors `anti_alias_activation_cuda.forward(inputs, up_ftr, down_ftr, alpha, beta)`
(alias_free_activation/cuda/anti_alias_activation.cpp:19-22) and the module
`alias_free_activation/cuda/activation1d.py:35-77` so `AMPBlock1`/`BigVGAN` built with
`use_cuda_kernel=True` work unmodified.  Same contract as the reference: returns a NEW
tensor (same dtype/device, no grad); contiguity is required; unsupported dtype ->
RuntimeError; launches on the current stream; backward is not implemented.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib


def kaiser_sinc_filter12(cutoff=0.25, half_width=0.3, kernel_size=12):
    """The 12 taps both resamplers use (filter.py:30-62; ratio 2 -> cutoff .25, half-width .3)."""
    import math

    half = kernel_size // 2
    A = 2.285 * (half - 1) * math.pi * (4 * half_width) + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = torch.arange(-half, half) + 0.5
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return (filt / filt.sum()).view(1, 1, kernel_size)


def forward(inputs, up_ftr, down_ftr, alpha, beta):
    """y = Down2(SnakeBeta(Up2(inputs))); inputs [B,C,T] fp32 on the GPU; alpha/beta log-scale [C]."""
    if inputs.dtype != torch.float32:
        raise RuntimeError(f"anti_alias_activation: unsupported dtype {inputs.dtype} (the IndexTTS2 pipeline runs BigVGAN in fp32, infer_v2.py:735)")
    if not inputs.is_cuda:
        raise RuntimeError("anti_alias_activation: HIP kernel needs a GPU tensor (no CPU fallback)")
    if inputs.dim() != 3:
        raise RuntimeError("anti_alias_activation: expected [B,C,T]")
    x = inputs.contiguous()
    B, Cc, T = x.shape
    up = up_ftr.to(device=x.device, dtype=torch.float32).contiguous().view(-1)
    down = down_ftr.to(device=x.device, dtype=torch.float32).contiguous().view(-1)
    a = alpha.to(device=x.device, dtype=torch.float32).contiguous()
    b = beta.to(device=x.device, dtype=torch.float32).contiguous()
    if up.numel() != 12 or down.numel() != 12 or a.numel() != Cc or b.numel() != Cc:
        raise RuntimeError("anti_alias_activation: filter must have 12 taps and alpha/beta one entry per channel")
    y = torch.empty_like(x)
    L = _lib.lib()
    rc = L.ixtts_aa_snake_f32(x.data_ptr(), y.data_ptr(), up.data_ptr(), down.data_ptr(), a.data_ptr(), b.data_ptr(),
                              B, Cc, T, _lib.current_stream_ptr())
    _lib.check(rc, "ixtts_aa_snake_f32")
    return y


class FusedAntiAliasActivation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, up_ftr, down_ftr, alpha, beta):
        return forward(inputs, up_ftr, down_ftr, alpha, beta)

    @staticmethod
    def backward(ctx, output_grads):
        raise NotImplementedError


class Activation1d(nn.Module):
    """Drop-in for `alias_free_activation.cuda.activation1d.Activation1d` (fused path only)."""

    def __init__(self, activation, up_ratio=2, down_ratio=2, up_kernel_size=12, down_kernel_size=12, fused=True):
        super().__init__()
        if (up_ratio, down_ratio, up_kernel_size, down_kernel_size) != (2, 2, 12, 12):
            raise NotImplementedError("the fused kernel hard-codes ratio 2 / 12 taps (activation1d.py:16-20)")
        self.act = activation
        self.fused = fused
        self.register_buffer("up_filter", kaiser_sinc_filter12())
        self.register_buffer("down_filter", kaiser_sinc_filter12())

    def forward(self, x):
        alpha = self.act.alpha.data
        beta = self.act.beta.data if hasattr(self.act, "beta") else alpha  # Snake shares alpha (activation1d.py:60-66)
        if not getattr(self.act, "alpha_logscale", True):
            alpha, beta = torch.log(alpha), torch.log(beta)
        return FusedAntiAliasActivation.apply(x, self.up_filter, self.down_filter, alpha, beta)
