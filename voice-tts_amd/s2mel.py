"""PyTorch glue for the semantic-to-mel stage -- `north_star`: "PyTorch-ROCm hosts the
tensors and the s2mel/vqvae glue".  Functional restatement (weights in a flat dict keyed like the reference's
state dicts, weight norm folded at load) of

  infer_v2.py:713-731                      gpt_layer -> vq2emb + latent -> length_regulator -> cat(prompt) -> CFM -> crop
  s2mel/modules/commons.py:388-438         MyModel (`models.{gpt_layer,length_regulator,cfm}`)
  s2mel/modules/length_regulator.py:28-141 InterpolateRegulator (continuous input, nearest interpolation)
  s2mel/modules/flow_matching.py:30-115    BASECFM.inference / solve_euler (CFG batch-2 stacking)
  s2mel/modules/diffusion_transformer.py:103-257  DiT (U-ViT skips, long skip, WaveNet head)
  s2mel/modules/gpt_fast/model.py:121-360  Transformer / AdaptiveLayerNorm(RMSNorm) / RoPE attention / SwiGLU
  s2mel/modules/wavenet.py:103-174 + encodec.py:192-228  WN with reflect-padded weight-normed convs
  utils/maskgct/.../factorized_vector_quantize.py:99-127  vq2emb = out_project(codebook[ids])

Runs on whatever device its tensors live on (GPU in the pipeline; CPU in the parity tests).  The CFM noise is an
explicit argument so CPU/GPU runs can share it.
"""
import math
import os

import torch
import torch.nn.functional as F

from .convs import conv1d  # GEMM forms: no MIOpen in the request path (convs.py)

from .weights import fold_weight_norm

S2MEL_CFG = dict(
    hidden_dim=512, num_heads=8, depth=13, in_channels=80, content_dim=512, style_dim=192,
    wavenet_hidden=512, wavenet_layers=8, wavenet_kernel=5, wavenet_dilation_rate=1,
    lr_channels=512, lr_in_channels=1024, lr_n_blocks=4, gpt_dim=1280, codebook_dim=8, codebook_size=8192, semantic_dim=1024,
)


def tiny_s2mel_cfg(**kw):
    c = dict(S2MEL_CFG)
    c.update(hidden_dim=64, num_heads=2, depth=5, content_dim=32, style_dim=12, wavenet_hidden=64, wavenet_layers=3,
             lr_channels=32, lr_in_channels=40, lr_n_blocks=2, gpt_dim=128, codebook_size=8194, semantic_dim=40)
    c.update(kw)
    return c


def _ffn_dim(dim):
    n_hidden = int(2 * (4 * dim) / 3)
    return n_hidden if n_hidden % 256 == 0 else n_hidden + 256 - (n_hidden % 256)


def s2mel_shapes(cfg=S2MEL_CFG):
    """(name, shape) of every tensor (weight norm already folded)."""
    H, Hw, C = cfg["hidden_dim"], cfg["wavenet_hidden"], cfg["in_channels"]
    assert H == Hw, "FinalLayer's adaLN takes the DiT timestep embedding: wavenet.hidden_dim must equal DiT.hidden_dim"
    out = [("gpt_layer.0.weight", (256, cfg["gpt_dim"])), ("gpt_layer.0.bias", (256,)), ("gpt_layer.1.weight", (128, 256)),
           ("gpt_layer.1.bias", (128,)), ("gpt_layer.2.weight", (cfg["semantic_dim"], 128)), ("gpt_layer.2.bias", (cfg["semantic_dim"],)),
           ("quantizer.codebook.weight", (cfg["codebook_size"], cfg["codebook_dim"])),
           ("quantizer.out_project.weight", (cfg["semantic_dim"], cfg["codebook_dim"], 1)), ("quantizer.out_project.bias", (cfg["semantic_dim"],))]
    lr = cfg["lr_channels"]
    out += [("length_regulator.content_in_proj.weight", (lr, cfg["lr_in_channels"])), ("length_regulator.content_in_proj.bias", (lr,))]
    for i in range(cfg["lr_n_blocks"]):
        out += [(f"length_regulator.model.{3 * i}.weight", (lr, lr, 3)), (f"length_regulator.model.{3 * i}.bias", (lr,)),
                (f"length_regulator.model.{3 * i + 1}.weight", (lr,)), (f"length_regulator.model.{3 * i + 1}.bias", (lr,))]
    n = 3 * cfg["lr_n_blocks"]
    out += [(f"length_regulator.model.{n}.weight", (lr, lr, 1)), (f"length_regulator.model.{n}.bias", (lr,))]
    e = "cfm.estimator."
    out += [(e + "t_embedder.mlp.0.weight", (H, 256)), (e + "t_embedder.mlp.0.bias", (H,)), (e + "t_embedder.mlp.2.weight", (H, H)), (e + "t_embedder.mlp.2.bias", (H,)),
            (e + "t_embedder2.mlp.0.weight", (Hw, 256)), (e + "t_embedder2.mlp.0.bias", (Hw,)), (e + "t_embedder2.mlp.2.weight", (Hw, Hw)), (e + "t_embedder2.mlp.2.bias", (Hw,)),
            (e + "cond_projection.weight", (H, cfg["content_dim"])), (e + "cond_projection.bias", (H,)),
            (e + "cond_x_merge_linear.weight", (H, H + 2 * C + cfg["style_dim"])), (e + "cond_x_merge_linear.bias", (H,)),
            (e + "skip_linear.weight", (H, H + C)), (e + "skip_linear.bias", (H,)),
            (e + "conv1.weight", (Hw, H)), (e + "conv1.bias", (Hw,)), (e + "conv2.weight", (C, Hw, 1)), (e + "conv2.bias", (C,)),
            (e + "res_projection.weight", (Hw, H)), (e + "res_projection.bias", (Hw,)),
            (e + "final_layer.linear.weight", (Hw, Hw)), (e + "final_layer.linear.bias", (Hw,)),
            (e + "final_layer.adaLN_modulation.1.weight", (2 * Hw, Hw)), (e + "final_layer.adaLN_modulation.1.bias", (2 * Hw,))]
    t = e + "transformer."
    inter = _ffn_dim(H)
    for i in range(cfg["depth"]):
        p = t + f"layers.{i}."
        out += [(p + "attention.wqkv.weight", (3 * H, H)), (p + "attention.wo.weight", (H, H)),
                (p + "feed_forward.w1.weight", (inter, H)), (p + "feed_forward.w3.weight", (inter, H)), (p + "feed_forward.w2.weight", (H, inter)),
                (p + "ffn_norm.project_layer.weight", (2 * H, H)), (p + "ffn_norm.project_layer.bias", (2 * H,)), (p + "ffn_norm.norm.weight", (H,)),
                (p + "attention_norm.project_layer.weight", (2 * H, H)), (p + "attention_norm.project_layer.bias", (2 * H,)), (p + "attention_norm.norm.weight", (H,)),
                (p + "skip_in_linear.weight", (H, 2 * H)), (p + "skip_in_linear.bias", (H,))]
    out += [(t + "norm.project_layer.weight", (2 * H, H)), (t + "norm.project_layer.bias", (2 * H,)), (t + "norm.norm.weight", (H,))]
    w = e + "wavenet."
    nl, k = cfg["wavenet_layers"], cfg["wavenet_kernel"]
    out += [(w + "cond_layer.conv.conv.weight", (2 * Hw * nl, Hw, 1)), (w + "cond_layer.conv.conv.bias", (2 * Hw * nl,))]
    for i in range(nl):
        rs = 2 * Hw if i < nl - 1 else Hw
        out += [(w + f"in_layers.{i}.conv.conv.weight", (2 * Hw, Hw, k)), (w + f"in_layers.{i}.conv.conv.bias", (2 * Hw,)),
                (w + f"res_skip_layers.{i}.conv.conv.weight", (rs, Hw, 1)), (w + f"res_skip_layers.{i}.conv.conv.bias", (rs,))]
    return out


def make_s2mel_weights(cfg=S2MEL_CFG, seed=1234):
    """Seeded synthetic weights (no checkpoint exists offline): N(0, 1/fan_in) matrices, small biases, gains near 1."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in s2mel_shapes(cfg):
        if name.endswith("norm.weight") or (".model." in name and len(shape) == 1 and name.endswith(".weight")):
            W[name] = 1.0 + 0.1 * torch.randn(*shape, generator=g)
        elif name.endswith(".bias"):
            W[name] = 0.02 * torch.randn(*shape, generator=g)
        elif name == "quantizer.codebook.weight":
            W[name] = torch.randn(*shape, generator=g)
        else:
            fan = 1
            for d in shape[1:]:
                fan *= d
            W[name] = torch.randn(*shape, generator=g) / math.sqrt(fan)
    return W


def _pad_reflect(x, left, right):
    """encodec.pad1d(mode='reflect'): zero-extend first when the input is shorter than the pad (encodec.py:96-113)."""
    L = x.shape[-1]
    extra = 0
    if L <= max(left, right):
        extra = max(left, right) - L + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (left, right), mode="reflect")
    return y[..., : y.shape[-1] - extra]


def attn_full(q, k, v, scale=None):
    """softmax(scale * q k^T) v for q, k, v [B, T, h, 64] fp32 CUDA tensors (any strides with d contiguous) through
    libixtts_hip.so's fp32-MFMA flash kernel (csrc/attn_full.hip); returns [B, T, h, 64] contiguous."""
    import ctypes as C

    from . import _lib

    B, T, Hh, d = q.shape
    assert d == 64 and q.is_cuda and q.dtype == torch.float32 and k.shape == q.shape and v.shape == q.shape
    for t in (q, k, v):
        assert t.stride(3) == 1 and t.stride() == q.stride() and t.stride(1) % 4 == 0 and t.data_ptr() % 16 == 0
    out = torch.empty(B, T, Hh, d, device=q.device, dtype=torch.float32)
    sc = float(scale if scale is not None else 1.0 / math.sqrt(d))
    L = _lib.lib()
    nws = L.ixtts_attn_full_workspace_bytes(B, Hh, T)
    ws = torch.empty(nws, device=q.device, dtype=torch.uint8)  # scratch for the key-split partials (caching allocator: no hipMalloc)
    with torch.cuda.device(q.device):
        rc = L.ixtts_attn_full_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, Hh, T, d, q.stride(0), q.stride(1),
                                   q.stride(2), out.stride(0), out.stride(1), out.stride(2), C.c_float(sc), ws.data_ptr(), nws,
                                   _lib.current_stream_ptr())
    _lib.check(rc, "ixtts_attn_full_f32")
    return out


def _lin(x, W, name):
    return F.linear(x, W[name + ".weight"], W.get(name + ".bias"))


# ---------------------------------------------------------------------------------- fused row / element ops
# On the GPU these are single HIP passes (csrc/dit_ops.hip, declared in include/ixtts_hip.h); on the CPU (parity tests
# against the reference's modules) the same arithmetic in torch.
def _hip_call(name, *args):
    from . import _lib

    _lib.check(getattr(_lib.lib(), name)(*args, _lib.current_stream_ptr()), name)


def adaln_rmsnorm(x, wb, g, eps=1e-5):
    """AdaptiveLayerNorm over RMSNorm (gpt_fast/model.py:18-37,362-372): x [B,T,H], wb = project_layer(c) [B,2H] (weight | bias)."""
    B, T, H = x.shape
    if x.is_cuda:
        import ctypes as C

        x, wb = x.contiguous(), wb.contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _hip_call("ixtts_adaln_rmsnorm_f32", x.data_ptr(), wb.data_ptr(), g.data_ptr(), out.data_ptr(), B, T, H, C.c_float(eps))
        return out
    w, b = wb[:, None, :H], wb[:, None, H:]
    return torch.addcmul(b, w, F.rms_norm(x, (H,), g, eps))


def ln_modulate(x, ss, eps=1e-6):
    """FinalLayer: layer_norm(x, no affine) * (1 + scale) + shift, ss = adaLN_modulation(c) [B,2H] = (shift | scale)."""
    B, T, H = x.shape
    if x.is_cuda:
        import ctypes as C

        x, ss = x.contiguous(), ss.contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _hip_call("ixtts_ln_modulate_f32", x.data_ptr(), ss.data_ptr(), out.data_ptr(), B, T, H, C.c_float(eps))
        return out
    return F.layer_norm(x, (H,), None, None, eps) * (1 + ss[:, None, H:]) + ss[:, None, :H]


def swiglu(u):
    """silu(u[..., :F]) * u[..., F:] on the fused [w1; w3] GEMM output (gpt_fast/model.py:316-326)."""
    Fd = u.shape[-1] // 2
    if u.is_cuda:
        u = u.contiguous()
        out = torch.empty(*u.shape[:-1], Fd, device=u.device, dtype=u.dtype)
        with torch.cuda.device(u.device):
            _hip_call("ixtts_swiglu_f32", u.data_ptr(), out.data_ptr(), u.numel() // (2 * Fd), Fd)
        return out
    return F.silu(u[..., :Fd]) * u[..., Fd:]


def wn_gate(a, g, off, C_):
    """tanh(a[:, :C] + g_a) * sigmoid(a[:, C:] + g_b) with (g_a | g_b) = g[:, off:off+2C] (wavenet.py:142-160)."""
    B, _, T = a.shape
    if a.is_cuda:
        a, g = a.contiguous(), g.contiguous()
        out = torch.empty(B, C_, T, device=a.device, dtype=a.dtype)
        with torch.cuda.device(a.device):
            _hip_call("ixtts_wn_gate_f32", a.data_ptr(), g.data_ptr(), out.data_ptr(), B, C_, T, g.shape[1], off)
        return out
    x = a + g[:, off:off + 2 * C_, None]
    return torch.tanh(x[:, :C_]) * torch.sigmoid(x[:, C_:])


def wn_gate_rows(a, g, off, C_, rows_per_batch):
    """`wn_gate` in row layout: a [rows, 2C] -> [rows, C]; row r uses the gate biases of batch entry min(r // rows_per_batch, B-1)."""
    rows, B = a.shape[0], g.shape[0]
    if a.is_cuda:
        a, g = a.contiguous(), g.contiguous()
        out = torch.empty(rows, C_, device=a.device, dtype=a.dtype)
        with torch.cuda.device(a.device):
            _hip_call("ixtts_wn_gate_rows_f32", a.data_ptr(), g.data_ptr(), out.data_ptr(), rows, C_, rows_per_batch, B, g.shape[1], off)
        return out
    b = (torch.arange(rows, device=a.device) // rows_per_batch).clamp(max=B - 1)
    x = a + g[b, off:off + 2 * C_]
    return torch.tanh(x[:, :C_]) * torch.sigmoid(x[:, C_:])


def reflect_halo_rows(p, T, left, right):
    """Refresh, in place, the reflect padding of p [B, left+T+right, C] from its interior rows."""
    B, _, C_ = p.shape
    if p.is_cuda:
        assert p.is_contiguous()
        with torch.cuda.device(p.device):
            _hip_call("ixtts_reflect_halo_rows_f32", p.data_ptr(), B, T, C_, left, right)
        return p
    if left:
        p[:, :left] = p[:, left + 1:2 * left + 1].flip(1)
    if right:
        p[:, left + T:] = p[:, left + T - 1 - right:left + T - 1].flip(1)
    return p


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunable", "gfx950_s2mel.csv")


def use_tuned_gemms(path=TUNED_GEMMS):
    """Library-GEMM selection for the DiT's fp32 GEMMs: PyTorch's TunableOp picks, per (layout, M, N, K), the fastest
    hipBLASLt / rocBLAS solution from a results file recorded on an MI355X with `tools/tune_gemms.py` (production shapes,
    T = 430 + 1892: wqkv 79 -> 56 us, [w1; w3] 136 -> 106 us, wo 40 -> 26 us).  Shapes without an entry keep the default
    heuristic; `IXTTS_TUNE=1` tunes new shapes online (seconds per shape, once).  Returns True when the file was taken."""
    if not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tn

    tn.enable(True)
    tn.tuning_enable(os.environ.get("IXTTS_TUNE") == "1")
    try:
        import tempfile

        tn.set_filename(os.path.join(tempfile.gettempdir(), "ixtts_tunableop_online.csv"))  # where online tuning (if on) records
    except Exception:
        pass
    try:
        return bool(os.path.isfile(path) and tn.read_file(path))
    except Exception:  # a file recorded under other library versions is refused by its validators: default heuristics
        return False


# The shipped results were recorded at ONE segment length (T = 430 prompt + 1892 target frames; the WaveNet head on Th = 1909 of them):
# TunableOp keys on the exact row count, so a segment of any other length fell back to the default heuristic -- 202 ms for 1806 frames
# against 196 ms for 2322 (tools/s2mel_scaling.py).  The winning solutions are Tensile kernels without a size-multiple requirement on
# the row dimension (`AF0EM1_AF1EM1` in their names): the same solution serves every row count, so the entries are re-keyed per length.
_TUNED_T0, _TUNED_TH0 = 2322, 1909
_tuned_lengths = set()


def extend_tuned_gemms(T, Th, path=TUNED_GEMMS):
    """Re-key the recorded winners of the DiT / WaveNet GEMMs (rows 2T, T, 2Th, 2Th + 4, Th at the recorded length) to this segment's
    row counts and hand them to TunableOp.  Once per distinct (T, Th); no-op when the results file was not taken."""
    if (T, Th) in _tuned_lengths or (T, Th) == (_TUNED_T0, _TUNED_TH0):
        return False
    _tuned_lengths.add((T, Th))
    import tempfile

    import torch.cuda.tunable as tn

    sub = {str(2 * _TUNED_T0): str(2 * T), str(_TUNED_T0): str(T), str(_TUNED_TH0): str(Th), str(2 * _TUNED_TH0): str(2 * Th),
           str(2 * _TUNED_TH0 + 4): str(2 * Th + 4)}
    head, rows = [], []
    for line in open(path).read().splitlines():
        f = line.split(",")
        if f[0] == "Validator":
            head.append(line)
        elif len(f) >= 4 and f[2] != "Default" and f[0].endswith("_TN"):
            # TN only: both operands K-contiguous with fixed leading dimensions, the row count is a plain free index.  In the TT
            # entries the frame count is a leading dimension and the vectorised one: their solutions carry an alignment requirement
            # (an odd T returned garbage through a re-keyed TT entry -- TunableOp does not look at the library's status)
            toks = f[1].split("_")
            if any(t in sub for t in toks):
                rows.append(",".join([f[0], "_".join(sub.get(t, t) for t in toks)] + f[2:]))
    if not rows:
        return False
    with tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False) as fh:
        fh.write("\n".join(head + rows) + "\n")
    try:
        return bool(tn.read_file(fh.name))
    except Exception:
        return False
    finally:
        os.unlink(fh.name)


class S2Mel:
    def __init__(self, W, cfg=S2MEL_CFG, device="cpu"):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.tuned_gemms = False
        if self.device.type == "cuda" and os.environ.get("IXTTS_NO_TUNED_GEMMS") != "1":
            self.tuned_gemms = use_tuned_gemms()
        W = fold_weight_norm(dict(W))
        self.W = {k: v.to(self.device, torch.float32) for k, v in W.items()}
        H = cfg["hidden_dim"]
        self.head_dim = H // cfg["num_heads"]
        half = 128
        self.t_freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half).to(self.device)
        self._rope = None
        # derived weights (exact re-arrangements: concatenations / column splits of the reference's matrices)
        W, C_, L = self.W, cfg["in_channels"], cfg["depth"]
        t = "cfm.estimator.transformer."
        self.w13 = [torch.cat([W[t + f"layers.{i}.feed_forward.w1.weight"], W[t + f"layers.{i}.feed_forward.w3.weight"]], 0).contiguous() for i in range(L)]
        self.skip_w = {i: (W[t + f"layers.{i}.skip_in_linear.weight"][:, :H].contiguous(), W[t + f"layers.{i}.skip_in_linear.weight"][:, H:].contiguous())
                       for i in range(L // 2 + 1, L)}
        names = [t + f"layers.{i}.{n}" for i in range(L) for n in ("attention_norm", "ffn_norm")] + [t + "norm"]
        self.proj_w = torch.cat([W[n + ".project_layer.weight"] for n in names], 0).contiguous()  # every AdaLN projection of a step: one GEMM
        self.proj_b = torch.cat([W[n + ".project_layer.bias"] for n in names], 0).contiguous()
        mw = W["cfm.estimator.cond_x_merge_linear.weight"]  # input = [x (C) | prompt_x (C) | cond (H) | style]
        self.merge_x, self.merge_rest = mw[:, :C_].contiguous(), mw[:, C_:].contiguous()
        wn = "cfm.estimator.wavenet."
        self.wn_taps = [[W[wn + f"in_layers.{i}.conv.conv.weight"][:, :, j].contiguous() for j in range(cfg["wavenet_kernel"])]
                        for i in range(cfg["wavenet_layers"])]
        # WaveNet with the residual / skip 1x1 convs as in-place accumulating GEMMs: their biases are constants per channel, so
        # the residual ones are carried forward into the next layer's conv bias (a reflect-padded constant stays a constant:
        # conv(x + c) = conv(x) + (sum of taps) c) and the skip ones are added once at the end
        nl, Hw = cfg["wavenet_layers"], cfg["wavenet_hidden"]
        self.wn_r1, self.wn_r2, self.wn_bin = [], [], []
        carry = torch.zeros(Hw, device=self.device)
        self.wn_out_bias = torch.zeros(Hw, device=self.device)
        for i in range(nl):
            rw, rb = W[wn + f"res_skip_layers.{i}.conv.conv.weight"][:, :, 0], W[wn + f"res_skip_layers.{i}.conv.conv.bias"]
            self.wn_bin.append(W[wn + f"in_layers.{i}.conv.conv.bias"] + sum(self.wn_taps[i]) @ carry)
            if i < nl - 1:
                self.wn_r1.append(rw[:Hw].contiguous())
                self.wn_r2.append(rw[Hw:].contiguous())
                carry = carry + rb[:Hw]
                self.wn_out_bias = self.wn_out_bias + rb[Hw:]
            else:
                self.wn_r1.append(None)
                self.wn_r2.append(rw.contiguous())
                self.wn_out_bias = self.wn_out_bias + rb
        sw = W["cfm.estimator.skip_linear.weight"]  # input = [x_res (H) | x (C)]
        self.skipl_res, self.skipl_x = sw[:, :H].contiguous(), sw[:, H:].contiguous()

    # ------------------------------------------------------------------ small pieces
    def gpt_layer(self, latent):
        x = _lin(latent, self.W, "gpt_layer.0")
        x = _lin(x, self.W, "gpt_layer.1")
        return _lin(x, self.W, "gpt_layer.2")

    def vq2emb(self, codes):
        """codes [B, n] -> [B, n, semantic_dim] (FactorizedVectorQuantize.vq2emb, then the pipeline's transpose)."""
        emb = self.W["quantizer.codebook.weight"][codes.long()]  # [B,n,8]
        out = conv1d(emb.transpose(1, 2), self.W["quantizer.out_project.weight"], self.W["quantizer.out_project.bias"])
        return out.transpose(1, 2)

    def length_regulator(self, x, ylens):
        """InterpolateRegulator.forward, continuous input: x [B, n, in] -> [B, max(ylens), C] (masked)."""
        W = self.W
        x = _lin(x, W, "length_regulator.content_in_proj")
        T = int(ylens.max())
        x = F.interpolate(x.transpose(1, 2).contiguous(), size=T, mode="nearest")
        for i in range(self.cfg["lr_n_blocks"]):
            x = conv1d(x, W[f"length_regulator.model.{3 * i}.weight"], W[f"length_regulator.model.{3 * i}.bias"], padding=1)
            x = F.group_norm(x, 1, W[f"length_regulator.model.{3 * i + 1}.weight"], W[f"length_regulator.model.{3 * i + 1}.bias"], 1e-5)
            x = F.mish(x)
        n = 3 * self.cfg["lr_n_blocks"]
        x = conv1d(x, W[f"length_regulator.model.{n}.weight"], W[f"length_regulator.model.{n}.bias"])
        mask = (torch.arange(T, device=x.device).unsqueeze(0) < ylens.unsqueeze(1)).unsqueeze(-1)
        return x.transpose(1, 2) * mask

    def _t_embed(self, t, prefix):
        args = 1000.0 * t[:, None].float() * self.t_freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        h = F.silu(_lin(emb, self.W, prefix + ".mlp.0"))
        return _lin(h, self.W, prefix + ".mlp.2")

    def _rope_cache(self, T):
        if self._rope is None or self._rope.shape[0] < T:
            n = self.head_dim
            freqs = 1.0 / (10000.0 ** (torch.arange(0, n, 2)[: n // 2].float() / n))
            ang = torch.outer(torch.arange(max(T, 64)).float(), freqs)
            self._rope = torch.polar(torch.ones_like(ang), ang).to(self.device)  # complex64 [T, d/2]
        return self._rope[:T]

    @staticmethod
    def _rotary(x, fc):
        """apply_rotary_emb (gpt_fast/model.py:348-360): (x0 + i x1) * (cos + i sin) on interleaved pairs, as one
        complex multiply.  x [B,T,h,d] fp32, fc complex [T,d/2]."""
        xc = torch.view_as_complex(x.reshape(*x.shape[:-1], -1, 2))
        return torch.view_as_real(xc * fc.view(1, fc.shape[0], 1, fc.shape[1])).flatten(3)

    def _attention(self, qkv, B, T, mask):
        """RoPE on q, k + softmax(q k^T / sqrt(d)) v (gpt_fast/model.py:289-312).  qkv [B*T, 3H] -> [B*T, H]."""
        H, nh, hd = self.cfg["hidden_dim"], self.cfg["num_heads"], self.head_dim
        fc = self._rope_cache(T)
        if mask is None and hd == 64 and qkv.is_cuda:
            # in-place rotation on the GEMM output, then the fp32-MFMA flash kernel on strided views of it: no copies
            with torch.cuda.device(qkv.device):
                _hip_call("ixtts_rope_qk_f32", qkv.data_ptr(), torch.view_as_real(fc).data_ptr(), B, T, H, hd)
            v4 = qkv.view(B, T, 3, nh, hd)
            return attn_full(v4[:, :, 0], v4[:, :, 1], v4[:, :, 2]).reshape(B * T, H)
        q, k, v = qkv.view(B, T, 3 * H).split([H, H, H], dim=-1)
        q = self._rotary(q.reshape(B, T, nh, hd), fc)
        k = self._rotary(k.reshape(B, T, nh, hd), fc)
        v = v.reshape(B, T, nh, hd)
        y = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask, dropout_p=0.0)
        return y.transpose(1, 2).reshape(B * T, H)

    def _transformer(self, x, c, mask):
        """Transformer.forward with U-ViT skips (gpt_fast/model.py:167-207): x [B,T,H], c [B,H] -> [B,T,H].  Residual adds
        ride in the GEMM epilogues (addmm), [w1; w3] is one GEMM, the skip concat is two accumulated GEMMs."""
        W, cfg = self.W, self.cfg
        B, T, H = x.shape
        L = cfg["depth"]
        # (weight | bias) of every AdaLN, one contiguous [B, 2H] slab per norm
        wb_all = F.linear(c, self.proj_w, self.proj_b).view(B, 2 * L + 1, 2 * H).transpose(0, 1).contiguous()
        x = x.reshape(B * T, H)
        skips = []
        for i in range(L):
            p = f"cfm.estimator.transformer.layers.{i}."
            if i > L // 2:
                wa, wb_ = self.skip_w[i]
                x = F.linear(skips.pop(), wb_, W[p + "skip_in_linear.bias"]).addmm_(x, wa.t())
            a = adaln_rmsnorm(x.view(B, T, H), wb_all[2 * i], W[p + "attention_norm.norm.weight"])
            y = self._attention(F.linear(a.view(B * T, H), W[p + "attention.wqkv.weight"]), B, T, mask)
            # residuals accumulate in place where the input is private (layer inputs 1..L/2 are also the saved U-ViT skips)
            h = torch.addmm(x, y, W[p + "attention.wo.weight"].t()) if 0 < i <= L // 2 else x.addmm_(y, W[p + "attention.wo.weight"].t())
            f = adaln_rmsnorm(h.view(B, T, H), wb_all[2 * i + 1], W[p + "ffn_norm.norm.weight"])
            x = h.addmm_(swiglu(F.linear(f.view(B * T, H), self.w13[i])), W[p + "feed_forward.w2.weight"].t())
            if i < L // 2:
                skips.append(x)
        return adaln_rmsnorm(x.view(B, T, H), wb_all[2 * L], W["cfm.estimator.transformer.norm.norm.weight"])

    def _wavenet(self, x, x_mask, g, full):
        """WN.forward: reflect-padded dilated convs, tanh*sigmoid gate, residual/skip split (wavenet.py:142-167).
        x [B,Hw,T]; g [B,Hw] (timestep embedding); `full`: every sequence spans T, so the mask multiplies are identities."""
        W, cfg = self.W, self.cfg
        Hw, nl, k = cfg["wavenet_hidden"], cfg["wavenet_layers"], cfg["wavenet_kernel"]
        p = "cfm.estimator.wavenet."
        g = F.linear(g, W[p + "cond_layer.conv.conv.weight"][:, :, 0], W[p + "cond_layer.conv.conv.bias"])  # 1x1 conv on a length-1 signal
        out = None
        for i in range(nl):
            d = cfg["wavenet_dilation_rate"] ** i
            tot = (k - 1) * d
            right = tot // 2
            xin = _pad_reflect(x, tot - right, right)  # SConv1d non-causal: left = total - total//2 (encodec.py:224-227)
            if x.is_cuda:
                # k accumulated GEMMs on shifted views of the padded input (one per tap) instead of im2col + GEMM
                wk, T_ = self.wn_taps[i], x.shape[-1]
                acc = W[p + f"in_layers.{i}.conv.conv.bias"][None, :, None].expand(x.shape[0], -1, T_)
                for j in range(k):  # the first tap materialises bias + W_0 x, the others accumulate in place (no copies)
                    wj, xj = wk[j].expand(x.shape[0], -1, -1), xin[:, :, j * d:j * d + T_]
                    acc = torch.baddbmm(acc, wj, xj) if j == 0 else acc.baddbmm_(wj, xj)
                xin = acc
            else:
                xin = conv1d(xin, W[p + f"in_layers.{i}.conv.conv.weight"], W[p + f"in_layers.{i}.conv.conv.bias"], dilation=d)
            acts = wn_gate(xin, g, i * 2 * Hw, Hw)
            rs = conv1d(acts, W[p + f"res_skip_layers.{i}.conv.conv.weight"], W[p + f"res_skip_layers.{i}.conv.conv.bias"])
            if i < nl - 1:
                x = x + rs[:, :Hw]
                if not full:
                    x = x * x_mask
                out = rs[:, Hw:] if out is None else out + rs[:, Hw:]
            else:
                out = rs if out is None else out + rs
        return out if full else out * x_mask

    def _wavenet_gemm(self, x, g):
        """The same network for the pipeline's case (GPU, every sequence spans T): every conv is an accumulating batched GEMM,
        no element-wise adds in between (see __init__); returns the skip sum WITHOUT `wn_out_bias` (the caller's linear adds it)."""
        W, cfg = self.W, self.cfg
        Hw, nl, k = cfg["wavenet_hidden"], cfg["wavenet_layers"], cfg["wavenet_kernel"]
        B, _, T_ = x.shape
        p = "cfm.estimator.wavenet."
        g = F.linear(g, W[p + "cond_layer.conv.conv.weight"][:, :, 0], W[p + "cond_layer.conv.conv.bias"])
        x = x.contiguous()
        out = None
        for i in range(nl):
            d = cfg["wavenet_dilation_rate"] ** i
            tot = (k - 1) * d
            right = tot // 2
            xin = _pad_reflect(x, tot - right, right)
            acc = self.wn_bin[i][None, :, None].expand(B, -1, T_)
            for j in range(k):
                wj, xj = self.wn_taps[i][j].expand(B, -1, -1), xin[:, :, j * d:j * d + T_]
                acc = torch.baddbmm(acc, wj, xj) if j == 0 else acc.baddbmm_(wj, xj)
            acts = wn_gate(acc, g, i * 2 * Hw, Hw)
            if i < nl - 1:
                x = x.baddbmm_(self.wn_r1[i].expand(B, -1, -1), acts)
            r2 = self.wn_r2[i].expand(B, -1, -1)
            out = torch.bmm(r2, acts) if out is None else out.baddbmm_(r2, acts)
        return out

    def _wavenet_rows(self, h, g):
        """The same network with sequences as ROWS (h [B,T,C], dilation 1): every conv is a row-major GEMM `X W^T` (the
        library's fast layout; the channel-major form above runs them at 0.6x the rate).  The B sequences sit in one
        reflect-padded buffer [B, T+k-1, C]; tap j of the k-tap conv is that buffer, flattened, shifted by j rows, so ONE
        GEMM per tap covers all sequences -- the k-1 rows that straddle two sequences are computed and never read (their
        residual update lands in padding rows, which are refreshed before the next layer reads them).
        Returns the skip sum [B,T,C] (a strided view) WITHOUT `wn_out_bias`."""
        W, cfg = self.W, self.cfg
        Hw, nl, k = cfg["wavenet_hidden"], cfg["wavenet_layers"], cfg["wavenet_kernel"]
        assert cfg["wavenet_dilation_rate"] == 1
        B, T_, _ = h.shape
        p = "cfm.estimator.wavenet."
        g = F.linear(g, W[p + "cond_layer.conv.conv.weight"][:, :, 0], W[p + "cond_layer.conv.conv.bias"])
        tot = k - 1
        right = tot // 2
        left = tot - right
        Tp = T_ + tot
        P = torch.empty(B, Tp, Hw, device=h.device, dtype=h.dtype)
        P[:, left:left + T_] = h
        P2 = P.view(B * Tp, Hw)
        M = B * Tp - tot
        centre = P2[left:left + M]  # the input row under output row r
        out = torch.empty(B * Tp, Hw, device=h.device, dtype=h.dtype)
        for i in range(nl):
            reflect_halo_rows(P, T_, left, right)
            acc = torch.addmm(self.wn_bin[i], P2[:M], self.wn_taps[i][0].t())
            for j in range(1, k):
                acc.addmm_(P2[j:j + M], self.wn_taps[i][j].t())
            acts = wn_gate_rows(acc, g, i * 2 * Hw, Hw, Tp)
            if i < nl - 1:
                centre.addmm_(acts, self.wn_r1[i].t())
            if i == 0:
                torch.mm(acts, self.wn_r2[i].t(), out=out[:M])
            else:
                out[:M].addmm_(acts, self.wn_r2[i].t())
        return out.view(B, Tp, Hw)[:, :T_]

    # ------------------------------------------------------------------ DiT + CFM
    def dit_prepare(self, prompt_x, x_lens, style, cond):
        """Everything of DiT.forward that does not depend on (x, t): computed once per CFM solve instead of once per
        Euler step.  prompt_x [B,80,T]; cond [B,T,content]; style [B,style_dim]."""
        W = self.W
        e = "cfm.estimator."
        B, _, T = prompt_x.shape
        cond = _lin(cond, W, e + "cond_projection")
        rest = torch.cat([prompt_x.transpose(1, 2), cond, style[:, None, :].expand(B, T, style.shape[-1])], dim=-1)
        base = F.linear(rest, self.merge_rest, W[e + "cond_x_merge_linear.bias"])  # cond_x_merge_linear minus its x columns
        x_mask = (torch.arange(T, device=prompt_x.device).unsqueeze(0) < x_lens.unsqueeze(1)).unsqueeze(1)  # [B,1,T]
        full = int(x_lens.min()) >= T
        if full:
            attn_mask = None  # the pipeline always runs one full-length sequence: an all-true key mask is no mask
        else:
            attn_mask = x_mask[:, None, :].expand(x_mask.shape[0], 1, T, T)
            if attn_mask.shape[0] != B:
                attn_mask = attn_mask.expand(B, 1, T, T)
        return dict(B=B, T=T, base=base, x_mask=x_mask.to(base.dtype), attn_mask=attn_mask, full=full)

    def dit_step(self, ctx, x, t, out_from=0):
        """The (x, t)-dependent part of DiT.forward (diffusion_transformer.py:186-257).  x [B,80,T] or [1,80,T] shared by
        the whole batch (the CFG stack feeds the same x to both branches); t [B].

        out_from > 0: the caller only uses frames [out_from, T) of the result (the CFM solver zeroes the prompt frames of x
        after every step).  After the transformer nothing mixes frames except the WaveNet's k-tap convs, so the head runs on
        frames [out_from - halo, T) only, halo = the WaveNet's receptive field; the returned tensor covers [out_from, T) and
        equals the full computation there."""
        W, cfg = self.W, self.cfg
        e = "cfm.estimator."
        B, T, base = ctx["B"], ctx["T"], ctx["base"]
        xt = x.transpose(1, 2)  # [Bx,T,80]
        t1 = self._t_embed(t, e + "t_embedder")
        x_in = base + F.linear(xt, self.merge_x)  # broadcasts a shared x over the batch
        x_res = self._transformer(x_in, t1, ctx["attn_mask"])
        halo = sum((cfg["wavenet_kernel"] - 1) * cfg["wavenet_dilation_rate"] ** i for i in range(cfg["wavenet_layers"])) // 2 + 1
        lo = max(0, out_from - halo) if (ctx["full"] and out_from > 0) else 0
        if lo > 0:
            x_res, xt = x_res[:, lo:], xt[:, lo:]
        Th = T - lo
        if self.tuned_gemms and os.environ.get("IXTTS_TUNED_ANY_LENGTH", "1") == "1":
            extend_tuned_gemms(T, Th)
        x_res = F.linear(x_res, self.skipl_res, W[e + "skip_linear.bias"]) + F.linear(xt, self.skipl_x)
        h = _lin(x_res, W, e + "conv1")
        t2 = self._t_embed(t, e + "t_embedder2")
        gemm_form = h.is_cuda and ctx["full"] and Th > 2 * cfg["wavenet_kernel"] * cfg["wavenet_dilation_rate"] ** (cfg["wavenet_layers"] - 1)
        if gemm_form and cfg["wavenet_dilation_rate"] == 1:
            h = self._wavenet_rows(h, t2) + F.linear(x_res, W[e + "res_projection.weight"], W[e + "res_projection.bias"] + self.wn_out_bias)
        elif gemm_form:
            h = self._wavenet_gemm(h.transpose(1, 2), t2).transpose(1, 2) + F.linear(x_res, W[e + "res_projection.weight"],
                                                                                     W[e + "res_projection.bias"] + self.wn_out_bias)
        else:
            mask = ctx["x_mask"][:, :, lo:] if lo > 0 else ctx["x_mask"]
            h = self._wavenet(h.transpose(1, 2), mask, t2, ctx["full"]).transpose(1, 2) + _lin(x_res, W, e + "res_projection")
        ss = _lin(F.silu(t1), W, e + "final_layer.adaLN_modulation.1")
        h = ln_modulate(h, ss)
        h = _lin(h, W, e + "final_layer.linear").transpose(1, 2)
        out = conv1d(h, W[e + "conv2.weight"], W[e + "conv2.bias"])
        return out[:, :, out_from - lo:] if out_from > 0 else out

    def dit(self, x, prompt_x, x_lens, t, style, cond):
        """DiT.forward, eval mode.  x, prompt_x [B,80,T]; cond [B,T,content]."""
        return self.dit_step(self.dit_prepare(prompt_x, x_lens, style, cond), x, t)

    @torch.no_grad()
    def cfm_inference(self, mu, x_lens, prompt, style, n_timesteps=25, inference_cfg_rate=0.7, noise=None, temperature=1.0):
        """BASECFM.inference + solve_euler (flow_matching.py:30-115).  mu [B,T,content]; prompt [B,80,Tp]; noise [B,80,T]."""
        B, T = mu.shape[0], mu.shape[1]
        assert B == 1, "CFM inference is batch-1 (as the reference)"
        z = (torch.randn(B, self.cfg["in_channels"], T, device=mu.device) if noise is None else noise.to(mu.device, torch.float32)) * temperature
        t_span = torch.linspace(0, 1, n_timesteps + 1, device=mu.device)
        x = z.clone()
        Tp = prompt.shape[-1]
        prompt_x = torch.zeros_like(x)
        prompt_x[..., :Tp] = prompt[..., :Tp]
        x[..., :Tp] = 0
        if inference_cfg_rate > 0:
            # batch-2 stacking of the conditional and the null branch (the reference supports B == 1 only:
            # its stacked_t has 2 entries whatever B is, flow_matching.py:88-93); x is shared by the two
            ctx = self.dit_prepare(torch.cat([prompt_x, torch.zeros_like(prompt_x)], 0), x_lens, torch.cat([style, torch.zeros_like(style)], 0),
                                   torch.cat([mu, torch.zeros_like(mu)], 0))
        else:
            ctx = self.dit_prepare(prompt_x, x_lens, style, mu)
        t = t_span[0]
        # the prompt frames of x are reset to zero after every step (flow_matching.py:112): the velocity there is never used
        skip = Tp if ctx["full"] else 0
        for step in range(1, len(t_span)):
            dt = t_span[step] - t_span[step - 1]
            if inference_cfg_rate > 0:
                d = self.dit_step(ctx, x, torch.stack([t, t]), out_from=skip)
                dphi = (1.0 + inference_cfg_rate) * d[0:1] - inference_cfg_rate * d[1:2]
            else:
                dphi = self.dit_step(ctx, x, t.expand(B), out_from=skip)
            if skip > 0:
                x[:, :, skip:] += dt * dphi
            else:
                x = x + dt * dphi
            t = t + dt
            x[:, :, :Tp] = 0
        return x

    # ------------------------------------------------------------------ the stage as infer_v2.py:713-731 calls it
    @torch.no_grad()
    def __call__(self, latent, codes, code_lens, prompt_condition, ref_mel, style, n_timesteps=25, inference_cfg_rate=0.7, noise=None):
        lat = self.gpt_layer(latent)
        S = self.vq2emb(codes) + lat
        target = (code_lens * 1.72).long()
        cond = self.length_regulator(S, target)
        cat = torch.cat([prompt_condition, cond], dim=1)
        lens = torch.tensor([cat.shape[1]], device=cat.device, dtype=torch.long)
        mel = self.cfm_inference(cat, lens, ref_mel, style, n_timesteps, inference_cfg_rate, noise)
        return mel[:, :, ref_mel.shape[-1]:]
