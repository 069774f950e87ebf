"""Seam 2: BigVGAN-v2 generator backed by the HIP library.

Mirrors `indextts/s2mel/modules/bigvgan/bigvgan.py:243-400`: constructed from the same
hyper-parameter mapping (`config.json`), fed the same generator state dict (with or
without weight norm), called as `bigvgan(mel_fp32[B,80,F]) -> wav[B,1,256F]`
(`indextts/infer_v2.py:735`).  `remove_weight_norm()` / `eval()` are accepted for
drop-in compatibility (the fold happens once in `load_state_dict`, row V5).
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .weights import BIGVGAN_CFG, fold_weight_norm


class BigVGAN(nn.Module):
    def __init__(self, h=None, use_cuda_kernel=True, max_frames=4096, fast_sin=False, device=None):
        super().__init__()
        h = dict(BIGVGAN_CFG if h is None else h)
        if not use_cuda_kernel:
            raise RuntimeError("this BigVGAN only has the HIP path (use_cuda_kernel=True); there is no torch fallback")
        if h.get("resblock", "1") != "1" or h.get("activation", "snakebeta") != "snakebeta" or not h.get("snake_logscale", True):
            raise NotImplementedError("HIP BigVGAN implements the shipped config: AMPBlock1 + log-scale SnakeBeta (config.json:9,20-21)")
        if h.get("use_tanh_at_final", False) or h.get("use_bias_at_final", False):
            raise NotImplementedError("HIP BigVGAN implements use_tanh_at_final=false / use_bias_at_final=false (config.json:17-18)")
        self.h = h
        self.device_ = torch.device(device if device is not None else "cuda:0")
        cfg = _lib.BigVGANCfg()
        cfg.num_mels = h["num_mels"]
        cfg.upsample_initial_channel = h["upsample_initial_channel"]
        rates, ks = list(h["upsample_rates"]), list(h["upsample_kernel_sizes"])
        cfg.n_stages = len(rates)
        for i, (u, k) in enumerate(zip(rates, ks)):
            cfg.upsample_rates[i] = u
            cfg.upsample_kernel_sizes[i] = k
        rk, rd = list(h["resblock_kernel_sizes"]), list(h["resblock_dilation_sizes"])
        cfg.n_resblock_kernels = len(rk)
        for j, (k, d) in enumerate(zip(rk, rd)):
            cfg.resblock_kernel_sizes[j] = k
            for m in range(3):
                cfg.resblock_dilations[j][m] = d[m]
        cfg.max_frames = max_frames
        cfg.fast_sin = int(bool(fast_sin))
        self._cfg = cfg
        self.total_up = 1
        for u in rates:
            self.total_up *= u
        self._h = C.c_void_p()
        with torch.cuda.device(self.device_):
            _lib.check(_lib.lib().ixtts_bigvgan_create(C.byref(self._h), C.byref(cfg)), "ixtts_bigvgan_create")
        self._loaded = False

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=True):
        sd = fold_weight_norm({k: v for k, v in sd.items() if not k.endswith("filter")})
        L = _lib.lib()
        with torch.cuda.device(self.device_):
            for name, t in sd.items():
                t = t.detach().to("cpu", torch.float32).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                rc = L.ixtts_bigvgan_set_tensor(self._h, name.encode(), t.data_ptr(), shape, t.dim())
                if rc == -5 and not strict:
                    continue
                _lib.check(rc, f"ixtts_bigvgan_set_tensor({name})")
            _lib.check(L.ixtts_bigvgan_finalize(self._h), "ixtts_bigvgan_finalize")
        self._loaded = True
        return self

    def arena(self):
        """(device pointer, bytes) of the packed weight arena -- for the RCCL broadcast at load."""
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().ixtts_bigvgan_arena(self._h, C.byref(p), C.byref(n)), "ixtts_bigvgan_arena")
        return p.value, n.value

    def adopt_arena(self):
        with torch.cuda.device(self.device_):
            _lib.check(_lib.lib().ixtts_bigvgan_adopt_arena(self._h), "ixtts_bigvgan_adopt_arena")
        self._loaded = True

    def remove_weight_norm(self):
        return self

    def flops(self, B, F):
        return _lib.lib().ixtts_bigvgan_flops(self._h, B, F)

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        if not self._loaded:
            raise RuntimeError("BigVGAN: weights not loaded")
        if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 3 or x.shape[1] != self.h["num_mels"]:
            raise RuntimeError(f"BigVGAN.forward expects a CUDA fp32 [B,{self.h['num_mels']},F] mel, got {tuple(x.shape)} {x.dtype} {x.device}")
        x = x.contiguous()
        B, _, F = x.shape
        wav = torch.empty(B, 1, F * self.total_up, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            rc = _lib.lib().ixtts_bigvgan_forward(self._h, x.data_ptr(), B, F, wav.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "ixtts_bigvgan_forward")
        return wav

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                _lib.lib().ixtts_bigvgan_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass
