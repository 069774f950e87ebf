"""Weight recipes and loaders for the hot path.

* `GPT_CFG` / `BIGVGAN_CFG`: production shapes (the reference's
  `models/IndexTTS/config.yaml` is not in its repo, `bigvgan/config.json:11-21` is).
* `make_gpt_weights` / `make_bigvgan_weights`: the seeded synthetic-weight recipe of
  SURVEY.md 8(d) (no checkpoints exist offline).  Keys follow the reference's
  `state_dict()` names so a real `gpt.pth` / BigVGAN generator drops in unchanged.
* `load_gpt_checkpoint` / `load_bigvgan_checkpoint`: the reference's file formats
  (`indextts/utils/checkpoint.py:25-34`; `bigvgan.py:413-492`), safe loaders only.
* `fold_weight_norm`: row V5 (`bigvgan.py:388-400`), done once at load.
"""
import math

import torch

GPT_CFG = dict(
    model_dim=1280, layers=24, heads=20, max_text_tokens=600, max_mel_tokens=1815,
    number_text_tokens=12000, number_mel_codes=8194, start_mel_token=8192, stop_mel_token=8193,
    start_text_token=0, stop_text_token=1,
)

BIGVGAN_CFG = dict(
    num_mels=80,
    upsample_rates=(4, 4, 2, 2, 2, 2),
    upsample_kernel_sizes=(8, 8, 4, 4, 4, 4),
    upsample_initial_channel=1536,
    resblock_kernel_sizes=(3, 7, 11),
    resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
)


def tiny_gpt_cfg(model_dim=128, layers=2, heads=2, **kw):
    c = dict(GPT_CFG)
    c.update(model_dim=model_dim, layers=layers, heads=heads, max_text_tokens=40, max_mel_tokens=80,
             number_text_tokens=200)
    c.update(kw)
    return c


def tiny_bigvgan_cfg(upsample_initial_channel=64, **kw):
    c = dict(BIGVGAN_CFG)
    c.update(upsample_initial_channel=upsample_initial_channel)
    c.update(kw)
    return c


def make_gpt_weights(cfg=GPT_CFG, seed=1234, head_scale=50.0, std=0.02):
    """GPT-2 init (N(0, 0.02), c_proj scaled by 1/sqrt(2L); transformers_gpt2.py:689-716).

    `mel_head.weight` is scaled by `head_scale` to widen greedy argmax margins
    (SURVEY.md section 7 "hard parts" (i)).  LayerNorm gains get a small seeded
    perturbation so that a dropped/duplicated norm is detectable.
    """
    g = torch.Generator().manual_seed(seed)
    D, L = cfg["model_dim"], cfg["layers"]
    V = cfg["number_mel_codes"]

    def n(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    W = {}
    for i in range(L):
        p = f"gpt.h.{i}."
        W[p + "ln_1.weight"] = 1.0 + n(D, s=0.05)
        W[p + "ln_1.bias"] = n(D, s=0.02)
        W[p + "attn.c_attn.weight"] = n(D, 3 * D)
        W[p + "attn.c_attn.bias"] = n(3 * D, s=0.01)
        W[p + "attn.c_proj.weight"] = n(D, D, s=std / math.sqrt(2 * L))
        W[p + "attn.c_proj.bias"] = n(D, s=0.01)
        W[p + "ln_2.weight"] = 1.0 + n(D, s=0.05)
        W[p + "ln_2.bias"] = n(D, s=0.02)
        W[p + "mlp.c_fc.weight"] = n(D, 4 * D)
        W[p + "mlp.c_fc.bias"] = n(4 * D, s=0.01)
        W[p + "mlp.c_proj.weight"] = n(4 * D, D, s=std / math.sqrt(2 * L))
        W[p + "mlp.c_proj.bias"] = n(D, s=0.01)
    W["gpt.ln_f.weight"] = 1.0 + n(D, s=0.05)
    W["gpt.ln_f.bias"] = n(D, s=0.02)
    W["final_norm.weight"] = 1.0 + n(D, s=0.05)
    W["final_norm.bias"] = n(D, s=0.02)
    W["mel_head.weight"] = n(V, D) * head_scale
    W["mel_head.bias"] = n(V, s=0.01)
    W["mel_embedding.weight"] = n(V, D)
    W["mel_pos_embedding.emb.weight"] = n(cfg["max_mel_tokens"] + 3, D)
    W["text_embedding.weight"] = n(cfg["number_text_tokens"] + 1, D)
    W["text_pos_embedding.emb.weight"] = n(cfg["max_text_tokens"] + 2, D)
    W["speed_emb.weight"] = n(2, D, s=0.02)
    return W


def bigvgan_shapes(cfg=BIGVGAN_CFG):
    """(name, shape) of every tensor in the weight-norm-folded generator state dict."""
    out = []
    c = cfg["upsample_initial_channel"]
    out += [("conv_pre.weight", (c, cfg["num_mels"], 7)), ("conv_pre.bias", (c,))]
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, ku) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        out += [(f"ups.{i}.0.weight", (c, c // 2, ku)), (f"ups.{i}.0.bias", (c // 2,))]
        c //= 2
        for j, k in enumerate(cfg["resblock_kernel_sizes"]):
            p = f"resblocks.{i * nk + j}"
            for m in range(3):
                out += [(f"{p}.convs1.{m}.weight", (c, c, k)), (f"{p}.convs1.{m}.bias", (c,))]
                out += [(f"{p}.convs2.{m}.weight", (c, c, k)), (f"{p}.convs2.{m}.bias", (c,))]
            for m in range(6):
                out += [(f"{p}.activations.{m}.act.alpha", (c,)), (f"{p}.activations.{m}.act.beta", (c,))]
    out += [("activation_post.act.alpha", (c,)), ("activation_post.act.beta", (c,)), ("conv_post.weight", (1, c, 7))]
    return out


def make_bigvgan_weights(cfg=BIGVGAN_CFG, seed=1234, snake_std=0.2):
    """Seeded synthetic generator weights.

    Conv weights ~ N(0, 1/fan_in) (variance-preserving; the reference's
    `init_weights` N(0, 0.01) makes the 109-layer output vanish, which would make an
    absolute waveform tolerance meaningless); biases N(0, 0.02); log-scale Snake
    alpha/beta ~ N(0, snake_std) (reference init is 0, `activations.py:96-98`).
    """
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in bigvgan_shapes(cfg):
        if name.endswith(".weight"):
            if name.startswith("ups."):
                cin, _, k = shape
                u = cfg["upsample_rates"][int(name.split(".")[1])]
                fan = cin * k / u
            else:
                _, cin, k = shape
                fan = cin * k
            W[name] = torch.randn(*shape, generator=g) * (1.0 / math.sqrt(fan))
        elif name.endswith(".bias"):
            W[name] = torch.randn(*shape, generator=g) * 0.02
        else:
            W[name] = torch.randn(*shape, generator=g) * snake_std
    # residual branches: damp conv2 so 9 residual adds per stage stay O(1)
    for name in list(W):
        if ".convs2." in name and name.endswith(".weight"):
            W[name] = W[name] * 0.5
    # keep most of the waveform inside the final clamp(-1, 1)
    W["conv_post.weight"] = W["conv_post.weight"] * 0.05
    return W


def fold_weight_norm(sd):
    """Row V5: w = g * v / ||v|| per output channel (dim 0), both torch spellings."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g") or k.endswith("parametrizations.weight.original0"):
            continue
        if k.endswith(".weight_v"):
            base = k[: -len(".weight_v")]
            gk = base + ".weight_g"
        elif k.endswith("parametrizations.weight.original1"):
            base = k[: -len(".parametrizations.weight.original1")]
            gk = base + ".parametrizations.weight.original0"
        else:
            out[k] = v
            continue
        v = v.float()
        g = sd[gk].float()
        norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
        out[base + ".weight"] = g * v / norm
    return out


def load_gpt_checkpoint(path):
    """`indextts/utils/checkpoint.py:25-34`: optional 'model' key; tensors only."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if "model" in sd:
        sd = sd["model"]
    out = {}
    for k, v in sd.items():
        k = k[7:] if k.startswith("module.") else k
        # UnifiedVoice.state_dict() also holds `inference_model.*` aliases of the same tensors
        if k.startswith("inference_model."):
            continue
        out[k] = v.float()
    return out


def load_bigvgan_checkpoint(path):
    """`bigvgan.py:413-492`: {"generator": state_dict}; weight norm folded here."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if "generator" in sd:
        sd = sd["generator"]
    return fold_weight_norm({k: v for k, v in sd.items()})
