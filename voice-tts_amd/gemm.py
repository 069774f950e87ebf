"""Row N1: the linear layers of the s2mel DiT / WaveNet on the HIP split-product GEMM (csrc/gemm_x6.hip) instead of the library's
fp32 GEMM.  `PackedLinear` holds a weight matrix split once into the bf16 planes the kernel streams; `split(x)` turns activation
rows into the same planes (one small pass; several GEMMs over the same input -- or over row-shifted windows of it, the taps of a
Conv1d -- share it); `linear(x, pl, ...)` = `F.linear` / `addmm_`.  Device tensors only (the CPU leg of the glue keeps torch's GEMM)."""
import ctypes as C

import torch

from . import _lib


class PackedLinear:
    """W [N, K] fp32 (K a multiple of 64) -> packed planes on W's device; `bias` [N] or None rides along."""

    __slots__ = ("N", "K", "packed", "bias")

    def __init__(self, weight, bias=None):
        assert weight.is_cuda and weight.dim() == 2 and weight.shape[1] % 64 == 0, tuple(weight.shape)
        w = weight.detach().to(torch.float32).contiguous()
        self.N, self.K = int(w.shape[0]), int(w.shape[1])
        nbytes = _lib.lib().ixtts_gemm_x6_packed_bytes(self.N, self.K)
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        with torch.cuda.device(w.device):
            _lib.check(_lib.lib().ixtts_gemm_x6_pack(w.data_ptr(), self.packed.data_ptr(), self.N, self.K, _lib.current_stream_ptr()), "ixtts_gemm_x6_pack")
            torch.cuda.current_stream().synchronize()  # `w` may be a temporary
        self.bias = None if bias is None else bias.detach().to(w.device, torch.float32).contiguous()


class Planes:
    """Activation rows [rows, K] as split planes (device buffer) -- what `linear` consumes."""

    __slots__ = ("buf", "rows", "K")

    def __init__(self, buf, rows, K):
        self.buf, self.rows, self.K = buf, rows, K


_POOL = {}  # (device, bytes) -> scratch buffers reused across calls: the planes of an input live until the next split of that size


def usable(x, K):
    """Whether `split` can take x [..., K] as it is: device fp32, 16-byte aligned rows, last dim contiguous, K % 64 == 0."""
    return x.is_cuda and x.dtype == torch.float32 and K % 64 == 0 and x.shape[-1] == K and x.stride(-1) == 1 and x.data_ptr() % 16 == 0 and \
        (x.dim() == 1 or (x.stride(-2) % 4 == 0 and x.stride(-2) >= K))


def split(x, slot=0):
    """x [..., K] fp32 (contiguous, or a 2-D view with one row stride) -> Planes.  `slot` names the scratch buffer: planes made with
    the same (size, slot) overwrite each other."""
    K = int(x.shape[-1])
    if x.dim() != 2:
        x = x.reshape(-1, K) if x.is_contiguous() else x.contiguous().reshape(-1, K)
    rows, lda = int(x.shape[0]), int(x.stride(0))
    L = _lib.lib()
    nbytes = (K // 16) * 6 * int(L.ixtts_gemm_x6_rows_padded(rows)) * 16
    key = (x.device, nbytes, slot)
    buf = _POOL.get(key)
    if buf is None:
        buf = _POOL[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.ixtts_gemm_x6_split(x.data_ptr(), lda, buf.data_ptr(), rows, K, _lib.current_stream_ptr()), "ixtts_gemm_x6_split")
    return Planes(buf, rows, K)


def linear(x, pl, out=None, accumulate=False, bias=True, tile=0, row0=0, rows=None, lead=None):
    """out[..., N] (+)= x[..., K] @ W^T (+ bias).  x: an fp32 device tensor (split here) or `Planes` (then rows [row0, row0 + rows) of
    it are the GEMM's A: a row-shifted window = one Conv1d tap).  out: fp32 tensor with contiguous rows, or None (allocated, shaped
    `lead + (N,)`).  accumulate=True adds to what `out` holds (torch's addmm_)."""
    K, N = pl.K, pl.N
    if not isinstance(x, Planes):
        lead = tuple(x.shape[:-1]) if lead is None else lead
        x = split(x)
    assert x.K == K, (x.K, K)
    M = (x.rows - row0) if rows is None else int(rows)
    if out is None:
        assert not accumulate
        out = torch.empty(*(lead if lead is not None else (M,)), N, dtype=torch.float32, device=pl.packed.device)
    assert out.shape[-1] == N and out.stride(-1) == 1 and out.numel() >= M * N
    ldc = int(out.stride(-2)) if out.dim() >= 2 else N
    b = pl.bias if (bias and pl.bias is not None) else None
    with torch.cuda.device(pl.packed.device):
        rc = _lib.lib().ixtts_gemm_x6_f32(x.buf.data_ptr(), x.rows, int(row0), pl.packed.data_ptr(), b.data_ptr() if b is not None else None, out.data_ptr(), ldc,
                                          M, N, K, 1 if accumulate else 0, int(tile), _lib.current_stream_ptr())
    _lib.check(rc, "ixtts_gemm_x6_f32")
    return out
