"""CPU restatement of the BigVGAN-v2 vocoder path (fp32).  TEST INFRASTRUCTURE ONLY.

Rows V0-V5 of SURVEY.md section 8(a).  Reference files (relative to the reference root):
  indextts/s2mel/modules/bigvgan/bigvgan.py                          (BigVGAN, AMPBlock1)
  indextts/s2mel/modules/bigvgan/activations.py:62-120                (SnakeBeta)
  indextts/s2mel/modules/bigvgan/alias_free_activation/torch/act.py:8-30
  indextts/s2mel/modules/bigvgan/alias_free_activation/torch/resample.py:10-58
  indextts/s2mel/modules/bigvgan/alias_free_activation/torch/filter.py:30-101

Weights are a flat dict keyed exactly like the reference's state_dict after
`remove_weight_norm()` (bigvgan.py:388-400): "conv_pre.weight", "ups.0.0.weight",
"resblocks.4.convs1.2.bias", "resblocks.4.activations.5.act.alpha", ...
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# bigvgan/config.json:11-21 (22 kHz / 80 band / 256x generator)
BIGVGAN_CFG = dict(
    num_mels=80,
    upsample_rates=(4, 4, 2, 2, 2, 2),
    upsample_kernel_sizes=(8, 8, 4, 4, 4, 4),
    upsample_initial_channel=1536,
    resblock_kernel_sizes=(3, 7, 11),
    resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
)


def kaiser_sinc_filter12(cutoff=0.25, half_width=0.3, kernel_size=12):
    """12-tap Kaiser-windowed sinc low-pass, normalised to sum 1 (filter.py:30-62).

    Up- and down-sampler use the same taps: cutoff 0.5/ratio, half_width 0.6/ratio,
    ratio 2 (resample.py:22-24, 47-52).  Returns float32 numpy [kernel_size].
    """
    half = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)  # float32, as the reference
    time = torch.arange(-half, half) + 0.5  # even kernel (filter.py:47-48)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.to(torch.float32).numpy().copy()


def aa_snake(x, log_alpha, log_beta, filt=None):
    """y = Down2(SnakeBeta(Up2(x))) for x [B,C,T] fp32 (act.py:24-30).

    Written in polyphase/gather form rather than pad+conv_transpose+crop:
      Up2   (resample.py:29-38): replicate-pad 5/5, 2*conv_transpose(stride 2), crop 15/15
              u[2m]   = 2 * sum_{a=0..5} f[11-2a] * x[clamp(m-3+a)]
              u[2m+1] = 2 * sum_{a=0..5} f[10-2a] * x[clamp(m-2+a)]
      Snake (activations.py:107-120, logscale): s = u + sin(u*e^alpha)^2 / (e^beta + 1e-9)
      Down2 (filter.py:92-101, pad_left 5 / pad_right 6, stride 2):
              y[t] = sum_{k=0..11} f[k] * s[clamp(2t+k-5, 0, 2T-1)]
    """
    if filt is None:
        filt = kaiser_sinc_filter12()
    f = torch.as_tensor(filt, dtype=torch.float32)
    x = torch.as_tensor(x, dtype=torch.float32)
    B, C, T = x.shape
    if T == 0:
        return x.clone()
    m = torch.arange(T)
    ue = torch.zeros_like(x)
    uo = torch.zeros_like(x)
    for a in range(6):
        ue = ue + f[11 - 2 * a] * x[..., (m - 3 + a).clamp(0, T - 1)]
        uo = uo + f[10 - 2 * a] * x[..., (m - 2 + a).clamp(0, T - 1)]
    u = torch.stack((2.0 * ue, 2.0 * uo), dim=-1).reshape(B, C, 2 * T)
    alpha = torch.exp(torch.as_tensor(log_alpha, dtype=torch.float32)).view(1, C, 1)
    beta = torch.exp(torch.as_tensor(log_beta, dtype=torch.float32)).view(1, C, 1)
    s = u + (1.0 / (beta + 1e-9)) * torch.sin(u * alpha) ** 2
    t = torch.arange(T)
    y = torch.zeros_like(x)
    for k in range(12):
        y = y + f[k] * s[..., (2 * t + k - 5).clamp(0, 2 * T - 1)]
    return y


def get_padding(k, d=1):
    """bigvgan/utils.py:52-53."""
    return int((k * d - d) / 2)


def fold_weight_norm(sd):
    """V5: fold g*v/||v|| once at load (bigvgan.py:388-400; torch weight_norm, dim=0).

    Accepts both the legacy (`weight_g`/`weight_v`) and the parametrized
    (`parametrizations.weight.original0/1`) spellings; returns a new dict with
    plain `.weight` entries.
    """
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g") or k.endswith("parametrizations.weight.original0"):
            continue
        if k.endswith(".weight_v"):
            base = k[: -len(".weight_v")]
            g = sd[base + ".weight_g"]
        elif k.endswith("parametrizations.weight.original1"):
            base = k[: -len(".parametrizations.weight.original1")]
            g = sd[base + ".parametrizations.weight.original0"]
        else:
            out[k] = v
            continue
        v = torch.as_tensor(v, dtype=torch.float32)
        g = torch.as_tensor(g, dtype=torch.float32)
        norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
        out[base + ".weight"] = g * v / norm
    return out


def amp_block1(x, W, prefix, k, dilations, filt):
    """AMPBlock1.forward (bigvgan.py:132-141)."""
    for j, d in enumerate(dilations):
        xt = aa_snake(x, W[f"{prefix}.activations.{2 * j}.act.alpha"], W[f"{prefix}.activations.{2 * j}.act.beta"], filt)
        xt = F.conv1d(xt, W[f"{prefix}.convs1.{j}.weight"], W[f"{prefix}.convs1.{j}.bias"], dilation=d, padding=get_padding(k, d))
        xt = aa_snake(xt, W[f"{prefix}.activations.{2 * j + 1}.act.alpha"], W[f"{prefix}.activations.{2 * j + 1}.act.beta"], filt)
        xt = F.conv1d(xt, W[f"{prefix}.convs2.{j}.weight"], W[f"{prefix}.convs2.{j}.bias"], dilation=1, padding=get_padding(k, 1))
        x = xt + x
    return x


def bigvgan_forward(mel, W, cfg=BIGVGAN_CFG, filt=None, return_stages=False):
    """BigVGAN.forward (bigvgan.py:360-386): mel [B,num_mels,F] fp32 -> wav [B,1,prod(rates)*F].

    use_tanh_at_final=false, use_bias_at_final=false (config.json:17-18) -> clamp(-1,1).
    """
    if filt is None:
        filt = kaiser_sinc_filter12()
    W = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in W.items()}
    x = torch.as_tensor(mel, dtype=torch.float32)
    stages = []
    x = F.conv1d(x, W["conv_pre.weight"], W["conv_pre.bias"], padding=3)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, ku) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(x, W[f"ups.{i}.0.weight"], W[f"ups.{i}.0.bias"], stride=u, padding=(ku - u) // 2)
        xs = None
        for j, (k, dil) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            r = amp_block1(x, W, f"resblocks.{i * nk + j}", k, dil, filt)
            xs = r if xs is None else xs + r
        x = xs / nk
        if return_stages:
            stages.append(x)
    x = aa_snake(x, W["activation_post.act.alpha"], W["activation_post.act.beta"], filt)
    x = F.conv1d(x, W["conv_post.weight"], W.get("conv_post.bias"), padding=3)
    x = torch.clamp(x, min=-1.0, max=1.0)
    if return_stages:
        return x, stages
    return x


def pcm16(wav):
    """infer_v2.py:740,772,781: clamp(32767*wav, +-32767) in fp32, then .type(int16) (truncation)."""
    w = torch.clamp(32767 * torch.as_tensor(wav, dtype=torch.float32), -32767.0, 32767.0)
    return w.to(torch.int16)


def bigvgan_flops(F_frames, cfg=BIGVGAN_CFG):
    """Algorithmic conv FLOPs for F mel frames."""
    fl = 2 * cfg["num_mels"] * cfg["upsample_initial_channel"] * 7 * F_frames
    T = F_frames
    c = cfg["upsample_initial_channel"]
    for u, ku in zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"]):
        T *= u
        fl += 2 * c * (c // 2) * ku * (T // u)
        c //= 2
        for k in cfg["resblock_kernel_sizes"]:
            fl += 6 * 2 * c * c * k * T
    fl += 2 * c * 7 * T
    return fl


def bigvgan_shapes(cfg=BIGVGAN_CFG):
    """(name, shape) for every tensor of the weight-norm-folded state dict."""
    out = []
    c = cfg["upsample_initial_channel"]
    out += [("conv_pre.weight", (c, cfg["num_mels"], 7)), ("conv_pre.bias", (c,))]
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, ku) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        out += [(f"ups.{i}.0.weight", (c, c // 2, ku)), (f"ups.{i}.0.bias", (c // 2,))]
        c //= 2
        for j, k in enumerate(cfg["resblock_kernel_sizes"]):
            p = f"resblocks.{i * nk + j}"
            for n in range(3):
                out += [(f"{p}.convs1.{n}.weight", (c, c, k)), (f"{p}.convs1.{n}.bias", (c,))]
                out += [(f"{p}.convs2.{n}.weight", (c, c, k)), (f"{p}.convs2.{n}.bias", (c,))]
            for n in range(6):
                out += [(f"{p}.activations.{n}.act.alpha", (c,)), (f"{p}.activations.{n}.act.beta", (c,))]
    out += [("activation_post.act.alpha", (c,)), ("activation_post.act.beta", (c,)), ("conv_post.weight", (1, c, 7))]
    return out
