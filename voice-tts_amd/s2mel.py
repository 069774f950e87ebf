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

import torch
import torch.nn.functional as F

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
        assert t.stride(3) == 1 and t.stride() == q.stride()
    out = torch.empty(B, T, Hh, d, device=q.device, dtype=torch.float32)
    sc = float(scale if scale is not None else 1.0 / math.sqrt(d))
    with torch.cuda.device(q.device):
        rc = _lib.lib().ixtts_attn_full_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, Hh, T, d, q.stride(0), q.stride(1),
                                            q.stride(2), out.stride(0), out.stride(1), out.stride(2), C.c_float(sc), _lib.current_stream_ptr())
    _lib.check(rc, "ixtts_attn_full_f32")
    return out


def _lin(x, W, name):
    return F.linear(x, W[name + ".weight"], W.get(name + ".bias"))


class S2Mel:
    def __init__(self, W, cfg=S2MEL_CFG, device="cpu"):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        W = fold_weight_norm(dict(W))
        self.W = {k: v.to(self.device, torch.float32) for k, v in W.items()}
        H = cfg["hidden_dim"]
        self.head_dim = H // cfg["num_heads"]
        half = 128
        self.t_freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half).to(self.device)
        self._rope = None

    # ------------------------------------------------------------------ small pieces
    def gpt_layer(self, latent):
        x = _lin(latent, self.W, "gpt_layer.0")
        x = _lin(x, self.W, "gpt_layer.1")
        return _lin(x, self.W, "gpt_layer.2")

    def vq2emb(self, codes):
        """codes [B, n] -> [B, n, semantic_dim] (FactorizedVectorQuantize.vq2emb, then the pipeline's transpose)."""
        emb = self.W["quantizer.codebook.weight"][codes.long()]  # [B,n,8]
        out = F.conv1d(emb.transpose(1, 2), self.W["quantizer.out_project.weight"], self.W["quantizer.out_project.bias"])
        return out.transpose(1, 2)

    def length_regulator(self, x, ylens):
        """InterpolateRegulator.forward, continuous input: x [B, n, in] -> [B, max(ylens), C] (masked)."""
        W = self.W
        x = _lin(x, W, "length_regulator.content_in_proj")
        T = int(ylens.max())
        x = F.interpolate(x.transpose(1, 2).contiguous(), size=T, mode="nearest")
        for i in range(self.cfg["lr_n_blocks"]):
            x = F.conv1d(x, W[f"length_regulator.model.{3 * i}.weight"], W[f"length_regulator.model.{3 * i}.bias"], padding=1)
            x = F.group_norm(x, 1, W[f"length_regulator.model.{3 * i + 1}.weight"], W[f"length_regulator.model.{3 * i + 1}.bias"], 1e-5)
            x = F.mish(x)
        n = 3 * self.cfg["lr_n_blocks"]
        x = F.conv1d(x, W[f"length_regulator.model.{n}.weight"], W[f"length_regulator.model.{n}.bias"])
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

    def _ada_norm(self, x, c, prefix):
        """AdaptiveLayerNorm(RMSNorm): weight * (rms(x) * g) + bias with (weight, bias) = project_layer(c)."""
        W = self.W
        wb = _lin(c, W, prefix + ".project_layer")
        H = x.shape[-1]
        w, b = wb[..., :H], wb[..., H:]
        n = F.rms_norm(x, (H,), W[prefix + ".norm.weight"], 1e-5)
        return torch.addcmul(b, w, n)

    def _transformer(self, x, c, mask):
        W, cfg = self.W, self.cfg
        B, T, H = x.shape
        nh, hd = cfg["num_heads"], self.head_dim
        fc = self._rope_cache(T)
        L = cfg["depth"]
        skips = []
        for i in range(L):
            p = f"cfm.estimator.transformer.layers.{i}."
            if i > L // 2:
                x = _lin(torch.cat([x, skips.pop()], dim=-1), W, p + "skip_in_linear")
            a = self._ada_norm(x, c, p + "attention_norm")
            q, k, v = F.linear(a, W[p + "attention.wqkv.weight"]).split([H, H, H], dim=-1)
            q = self._rotary(q.view(B, T, nh, hd), fc)
            k = self._rotary(k.view(B, T, nh, hd), fc)
            v = v.reshape(B, T, nh, hd)
            if mask is None and hd == 64 and x.is_cuda:
                y = attn_full(q.contiguous(), k.contiguous(), v.contiguous()).reshape(B, T, H)  # HIP fp32-MFMA flash kernel
            else:
                y = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask, dropout_p=0.0)
                y = y.transpose(1, 2).reshape(B, T, H)
            h = x + F.linear(y, W[p + "attention.wo.weight"])
            f = self._ada_norm(h, c, p + "ffn_norm")
            x = h + F.linear(F.silu(F.linear(f, W[p + "feed_forward.w1.weight"])) * F.linear(f, W[p + "feed_forward.w3.weight"]), W[p + "feed_forward.w2.weight"])
            if i < L // 2:
                skips.append(x)
        return self._ada_norm(x, c, "cfm.estimator.transformer.norm")

    def _wavenet(self, x, x_mask, g):
        """WN.forward: reflect-padded dilated convs, tanh*sigmoid gate, residual/skip split (wavenet.py:142-167)."""
        W, cfg = self.W, self.cfg
        Hw, nl, k = cfg["wavenet_hidden"], cfg["wavenet_layers"], cfg["wavenet_kernel"]
        p = "cfm.estimator.wavenet."
        g = F.conv1d(g, W[p + "cond_layer.conv.conv.weight"], W[p + "cond_layer.conv.conv.bias"])
        out = torch.zeros_like(x)
        for i in range(nl):
            d = cfg["wavenet_dilation_rate"] ** i
            tot = (k - 1) * d
            right = tot // 2
            xin = _pad_reflect(x, tot - right, right)  # SConv1d non-causal: left = total - total//2 (encodec.py:224-227)
            xin = F.conv1d(xin, W[p + f"in_layers.{i}.conv.conv.weight"], W[p + f"in_layers.{i}.conv.conv.bias"], dilation=d)
            a = xin + g[:, i * 2 * Hw:(i + 1) * 2 * Hw, :]
            acts = torch.tanh(a[:, :Hw]) * torch.sigmoid(a[:, Hw:])
            rs = F.conv1d(acts, W[p + f"res_skip_layers.{i}.conv.conv.weight"], W[p + f"res_skip_layers.{i}.conv.conv.bias"])
            if i < nl - 1:
                x = (x + rs[:, :Hw]) * x_mask
                out = out + rs[:, Hw:]
            else:
                out = out + rs
        return out * x_mask

    # ------------------------------------------------------------------ DiT + CFM
    def dit(self, x, prompt_x, x_lens, t, style, cond):
        """DiT.forward, eval mode (diffusion_transformer.py:186-257).  x, prompt_x [B,80,T]; cond [B,T,content]."""
        W = self.W
        e = "cfm.estimator."
        B, _, T = x.shape
        t1 = self._t_embed(t, e + "t_embedder")
        cond = _lin(cond, W, e + "cond_projection")
        xt = x.transpose(1, 2)
        x_in = torch.cat([xt, prompt_x.transpose(1, 2), cond, style[:, None, :].repeat(1, T, 1)], dim=-1)
        x_in = _lin(x_in, W, e + "cond_x_merge_linear")
        x_mask = (torch.arange(T, device=x.device).unsqueeze(0) < x_lens.unsqueeze(1)).unsqueeze(1)  # [B,1,T]
        if int(x_lens.min()) >= T:
            attn_mask = None  # the pipeline always runs one full-length sequence: an all-true key mask is no mask
        else:
            attn_mask = x_mask[:, None, :].expand(x_mask.shape[0], 1, T, T)
            if attn_mask.shape[0] != B:
                attn_mask = attn_mask.expand(B, 1, T, T)
        x_res = self._transformer(x_in, t1.unsqueeze(1), attn_mask)
        x_res = _lin(torch.cat([x_res, xt], dim=-1), W, e + "skip_linear")
        h = _lin(x_res, W, e + "conv1").transpose(1, 2)
        t2 = self._t_embed(t, e + "t_embedder2")
        h = self._wavenet(h, x_mask.to(h.dtype), t2.unsqueeze(2)).transpose(1, 2) + _lin(x_res, W, e + "res_projection")
        ss = _lin(F.silu(t1), W, e + "final_layer.adaLN_modulation.1")
        Hw = self.cfg["wavenet_hidden"]
        shift, scale = ss[:, :Hw], ss[:, Hw:]
        h = F.layer_norm(h, (Hw,), None, None, 1e-6) * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)
        h = _lin(h, W, e + "final_layer.linear").transpose(1, 2)
        return F.conv1d(h, W[e + "conv2.weight"], W[e + "conv2.bias"])

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
        t = t_span[0]
        for step in range(1, len(t_span)):
            dt = t_span[step] - t_span[step - 1]
            if inference_cfg_rate > 0:
                # batch-2 stacking of the conditional and the null branch (the reference supports B == 1 only:
                # its stacked_t has 2 entries whatever B is, flow_matching.py:88-93)
                d = self.dit(torch.cat([x, x], 0), torch.cat([prompt_x, torch.zeros_like(prompt_x)], 0), x_lens, torch.stack([t, t]),
                             torch.cat([style, torch.zeros_like(style)], 0), torch.cat([mu, torch.zeros_like(mu)], 0))
                dphi, cfg_dphi = d.chunk(2, dim=0)
                dphi = (1.0 + inference_cfg_rate) * dphi - inference_cfg_rate * cfg_dphi
            else:
                dphi = self.dit(x, prompt_x, x_lens, t.expand(B), style, mu)
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
