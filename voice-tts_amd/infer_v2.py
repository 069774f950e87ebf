"""Host-side mirror of `indextts.infer_v2.IndexTTS2` for the hot path this repo builds.

Same constructor and `infer(...)` signature, return values, wav format and log lines as the
reference (`indextts/infer_v2.py:36-45,438-461,463-783`).  The two hot stages -- the GPT
(`inference_speech` + latent `forward`) and BigVGAN -- run in libixtts_hip.so; the stages
`north_star` leaves to PyTorch glue run on the same device through this package's torch
restatements: conformer/perceiver conditioners (`conditioning.py`), s2mel length regulator + CFM
(`s2mel.py`), and the once-per-prompt stages of infer_v2.py:508-580 (`prompt.py`: audio decode of
the five prompt forms, resampling, w2v-bert features through the installed `transformers`
classes, semantic codec `quantize`, reference mel, kaldi fbank + CAM++, emotion-matrix mix).

`IndexTTS2(cfg_path, model_dir).infer(wav, text)` therefore runs from the files of `model_dir`
alone (layout in INTEGRATION.md; nothing is ever downloaded); every component can also be handed
in as tensors (`*_state_dict=`, `w2v_bert=`, `emo_matrix=` ...), which is how the tests and the
bench build synthetic twins.  A `glue=` object still overrides any stage (see `Glue`).  There is
no silent fallback: a stage whose weights are missing makes `infer()` raise, naming the file.
"""
import logging
import os
import time
import warnings
import wave

import numpy as np
import torch

from .bigvgan import BigVGAN
from .gpt_engine import GptEngine
from .pipeline import prepare_gpt_inputs
from .weights import BIGVGAN_CFG, GPT_CFG, load_bigvgan_checkpoint, load_gpt_checkpoint

logger = logging.getLogger("indextts.infer_v2")


def _same_prompt(a, b):
    """The reference's cache test `cache_spk_audio_prompt != spk_audio_prompt` (infer_v2.py:508,566) for every prompt form:
    equal paths / bytes hit the cache even when they are distinct objects; arrays and tensors compare by content."""
    if a is b:
        return True
    if type(a) is not type(b):
        return False
    if isinstance(a, tuple):
        return len(a) == len(b) and all(_same_prompt(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return a.shape == b.shape and bool(np.array_equal(a, b))
    if isinstance(a, torch.Tensor):
        return a.shape == b.shape and bool(torch.equal(a, b))
    try:
        return bool(a == b)
    except Exception:
        return False


def load_s2mel_checkpoint(path):
    """`load_checkpoint2` (s2mel/modules/commons.py:568-624): {"net": {module: state_dict}}, DDP "module." prefixes stripped;
    flattened to `module.key` as `S2Mel` reads them.  Tensors only (weights_only)."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    net = sd["net"] if "net" in sd else sd
    out = {}
    for mod, params in net.items():
        if not isinstance(params, dict):
            continue
        for k, v in params.items():
            k = k[7:] if k.startswith("module.") else k
            if torch.is_tensor(v) and v.is_floating_point():
                out[f"{mod}.{k}"] = v.float()
    return out


def _s2mel_cfg_from_yaml(y, base):
    """`cfg.s2mel` (Appendix A layout) -> the keys `S2Mel` reads."""
    c = dict(base)
    d, w, lr = y.get("DiT") or {}, y.get("wavenet") or {}, y.get("length_regulator") or {}
    for src, dst in (("hidden_dim", "hidden_dim"), ("num_heads", "num_heads"), ("depth", "depth"), ("in_channels", "in_channels"), ("content_dim", "content_dim")):
        if src in d:
            c[dst] = d[src]
    for src, dst in (("hidden_dim", "wavenet_hidden"), ("num_layers", "wavenet_layers"), ("kernel_size", "wavenet_kernel"), ("dilation_rate", "wavenet_dilation_rate")):
        if src in w:
            c[dst] = w[src]
    if "channels" in lr:
        c["lr_channels"] = lr["channels"]
    if "in_channels" in lr:
        c["lr_in_channels"] = lr["in_channels"]
    if "sampling_ratios" in lr:
        c["lr_n_blocks"] = len(lr["sampling_ratios"])
    if "dim" in (y.get("style_encoder") or {}):
        c["style_dim"] = y["style_encoder"]["dim"]
    return c


class Glue:
    """Optional override of the PyTorch-hosted stages: a method that is implemented
    replaces the built-in stage, one that raises NotImplementedError leaves the built-in in place.  All tensors on `device`."""

    def tokenize(self, text, max_text_tokens_per_segment, quick_streaming_tokens=0):
        """-> list of segments, each a list of text token ids (front.py:313-327,345-436; infer_v2.py:582-583,617)."""
        raise NotImplementedError

    def speaker(self, spk_audio_prompt):
        """-> dict(spk_cond_emb [1,T,1024], style [1,192], prompt_condition [1,Tr,512], ref_mel [1,80,Tr]) (infer_v2.py:508-545)."""
        raise NotImplementedError

    def emotion(self, emo_audio_prompt):
        """-> emo_cond_emb [1,T,1024] (infer_v2.py:565-580)."""
        raise NotImplementedError

    def emo_vector_mix(self, emo_vector, style, use_random):
        """-> (emovec_mat [1,D], weight_sum) (infer_v2.py:552-563)."""
        raise NotImplementedError

    def merge_emovec(self, spk_cond_emb, emo_cond_emb, alpha):
        """UnifiedVoice.merge_emovec -> [1,D] (model_v2.py:742-747).  Only called when the GPT checkpoint lacks the encoders."""
        raise NotImplementedError

    def get_conditioning(self, spk_cond_emb):
        """UnifiedVoice.get_conditioning -> [32,D] (model_v2.py:514-543,684).  Only called when the GPT checkpoint lacks the encoders."""
        raise NotImplementedError

    def s2mel(self, latent, codes, code_lens, speaker):
        """gpt_layer + vq2emb + length_regulator + cfm.inference -> mel [1,80,F] after the prompt (infer_v2.py:713-731).
        Only called when no `s2mel_state_dict` was given."""
        raise NotImplementedError


class IndexTTS2:
    returns_pcm_without_path = True  # infer(output_path=None) -> (22050, int16 [N, 1]) (infer_v2.py:776-783): the server skips the temp file

    def __init__(self, cfg_path="models/IndexTTS/config.yaml", model_dir="models/IndexTTS", use_fp16=False, device=None,
                 use_cuda_kernel=None, use_deepspeed=False, *, glue=None, gpt_state_dict=None, bigvgan_state_dict=None,
                 s2mel_state_dict=None, gpt_cfg=None, bigvgan_cfg=None, cond_cfg=None, s2mel_cfg=None, tokenizer=None,
                 max_seq=2048, max_frames=4096, w2v_bert=None, w2v_stats=None, semantic_codec_state_dict=None, codec_cfg=None,
                 campplus_state_dict=None, emo_matrix=None, spk_matrix=None, emo_num=None, weight_broadcast=None):
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("the HIP hot path needs a GPU (no CPU fallback); pass device='cuda:N'")
            device = "cuda:0"
        self.device = torch.device(device)
        self.use_fp16 = bool(use_fp16)
        self.use_cuda_kernel = True if use_cuda_kernel is None else bool(use_cuda_kernel)
        if use_deepspeed:
            logger.info("use_deepspeed is ignored: the HIP decode engine stands in DeepSpeed's seam (model_v2.py:433-446)")
        self.model_dir = model_dir
        self.glue = glue
        gcfg, bcfg = dict(GPT_CFG if gpt_cfg is None else gpt_cfg), dict(BIGVGAN_CFG if bigvgan_cfg is None else bigvgan_cfg)
        cfg = {}
        if cfg_path and os.path.isfile(cfg_path):
            import yaml

            cfg = yaml.safe_load(open(cfg_path)) or {}
            for k in gcfg:
                if k in cfg.get("gpt", {}):
                    gcfg[k] = cfg["gpt"][k]
        self.cfg = cfg
        self.model_version = cfg.get("version")
        self.stop_mel_token = gcfg["stop_mel_token"]
        # reference precision: fp16 GPT under use_fp16 (infer_v2.py:79,88-89); here bf16 is the throughput mode
        self.gpt = GptEngine(gcfg, dtype="bf16" if use_fp16 else "f32", max_seq=max_seq, max_batch=3, device=self.device)
        # ---- weights: rank 0 (or a lone worker) reads model_dir; with `weight_broadcast=(rank, world)` (torch.distributed initialised,
        # backend nccl = RCCL over xGMI) the other workers of the node read nothing but config.yaml / bpe.model / the w2v-bert
        # directory: the glue tensors arrive as one packed message, the GPT and BigVGAN weights as their packed device arenas
        # (north_star: "RCCL over xGMI only for weight broadcast at load"; the reference's workers each read the files, server.py:28-77)
        self._bc = weight_broadcast
        peer = weight_broadcast is not None and int(weight_broadcast[0]) != 0
        files = dict(gpt=gpt_state_dict, bigvgan=bigvgan_state_dict, s2mel=s2mel_state_dict, codec=semantic_codec_state_dict, campplus=campplus_state_dict,
                     emo_matrix=emo_matrix, spk_matrix=spk_matrix)
        if not peer:
            files, bcfg = self._read_model_dir(files, cfg, model_dir, bcfg, bigvgan_cfg is None)
        if weight_broadcast is not None:
            files, bcfg = self._exchange_glue(files, bcfg, peer)
        gpt_state_dict, bigvgan_state_dict, s2mel_state_dict = files["gpt"], files["bigvgan"], files["s2mel"]
        semantic_codec_state_dict, campplus_state_dict, emo_matrix, spk_matrix = files["codec"], files["campplus"], files["emo_matrix"], files["spk_matrix"]
        if gpt_state_dict is None:
            raise FileNotFoundError("no GPT weights: pass gpt_state_dict=... or provide model_dir/gpt_checkpoint (checkpoint.py:25-34)")
        if bigvgan_state_dict is None:
            raise FileNotFoundError("no BigVGAN weights: pass bigvgan_state_dict=... or provide model_dir/bigvgan_generator.pt")
        if bigvgan_cfg is None and "conv_pre.weight" in bigvgan_state_dict:
            bcfg["upsample_initial_channel"] = int(bigvgan_state_dict["conv_pre.weight"].shape[0])  # the one width the tensors fix
        self.bigvgan = BigVGAN(bcfg, use_cuda_kernel=True, max_frames=max_frames, device=self.device)
        if not peer:
            self.gpt.load_state_dict(gpt_state_dict)
            self.bigvgan.load_state_dict(bigvgan_state_dict)
        if weight_broadcast is not None:  # the packed device arenas: one RCCL broadcast each, then `adopt_arena` on the receivers
            from . import sharding
            from .pipeline import arena_tensor

            gp, gn = self.gpt.arena()
            bp, bn = self.bigvgan.arena()
            sharding.broadcast_weights([arena_tensor(gp, gn, self.device), arena_tensor(bp, bn, self.device)], src=0)
            torch.cuda.synchronize(self.device)
            if peer:
                self.gpt.adopt_arena()
                self.bigvgan.adopt_arena()
        self._engines = {}  # further engine shapes over the same device weights, by slot count (infer_many, beam groups)
        D = gcfg["model_dim"]
        self.text_embedding = gpt_state_dict["text_embedding.weight"].to(self.device, torch.float32)
        self.text_pos_embedding = gpt_state_dict["text_pos_embedding.emb.weight"].to(self.device, torch.float32)
        self.speed_emb = gpt_state_dict["speed_emb.weight"].to(self.device, torch.float32)
        self.gpt_cfg = gcfg
        self.model_dim = D
        # rows N2 / N1 as PyTorch-ROCm glue, when their weights are there
        self.cond = None
        if "conditioning_encoder.embed.conv.0.weight" in gpt_state_dict:
            from .conditioning import COND_CFG, Conditioning

            if cond_cfg is None:
                cond_cfg = dict(COND_CFG, model_dim=D)
                for k in ("condition_module", "emo_condition_module"):
                    if k in cfg.get("gpt", {}):
                        cond_cfg[k] = {**cond_cfg[k], **{a: b for a, b in cfg["gpt"][k].items() if a in cond_cfg[k]}}
                # widths the yaml does not spell out are read off the tensors
                # Conv2dSubsampling2.out is Linear(D * ((idim - 1) // 2), D) (subsampling.py:152-153): only (idim - 1) // 2 enters the
                # arithmetic; the even idim with that quotient is the feature size of w2v-bert (1024 -> 511)
                eo = gpt_state_dict["conditioning_encoder.embed.out.0.weight"]
                cond_cfg["input_size"] = cfg.get("gpt", {}).get("input_size", 2 * (eo.shape[1] // eo.shape[0]) + 2)
                cond_cfg["perceiver_dim_head"] = gpt_state_dict["perceiver_encoder.layers.0.0.to_q.weight"].shape[0] // cond_cfg["condition_module"]["attention_heads"]
                cond_cfg["perceiver_depth"] = len({k.split(".")[2] for k in gpt_state_dict if k.startswith("perceiver_encoder.layers.")})
                cond_cfg["cnn_kernel"] = gpt_state_dict["conditioning_encoder.encoders.0.conv_module.depthwise_conv.weight"].shape[-1]
                cond_cfg["emo_dim"] = gpt_state_dict["emovec_layer.weight"].shape[1]
                cond_cfg["cond_num"] = gpt_state_dict["perceiver_encoder.latents"].shape[0]
            self.cond = Conditioning(gpt_state_dict, cond_cfg, device=self.device)
        # ---- prompt-side models (prompt.py; infer_v2.py:114-152,168-188): given tensors, else the files of model_dir
        from . import prompt as PR

        missing = []
        have = lambda *parts: os.path.exists(os.path.join(model_dir, *parts))
        codec_cfg = dict(PR.CODEC_CFG if codec_cfg is None else codec_cfg)
        for k in codec_cfg:
            if k in (cfg.get("semantic_codec") or {}):
                codec_cfg[k] = cfg["semantic_codec"][k]
        self.semantic_codec = PR.SemanticCodec(semantic_codec_state_dict, codec_cfg, self.device) if semantic_codec_state_dict is not None else None
        if self.semantic_codec is None:
            missing.append("semantic_codec/model.safetensors")
        self.s2mel = None
        if s2mel_state_dict is not None:
            from .s2mel import S2MEL_CFG, S2Mel

            scfg = dict(S2MEL_CFG if s2mel_cfg is None else s2mel_cfg)
            if s2mel_cfg is None and cfg.get("s2mel"):
                scfg = _s2mel_cfg_from_yaml(cfg["s2mel"], scfg)
            s2mel_state_dict = dict(s2mel_state_dict)
            if "quantizer.codebook.weight" not in s2mel_state_dict and self.semantic_codec is not None:
                # `semantic_codec.quantizer.vq2emb` (infer_v2.py:714) lives in the codec checkpoint, not in s2mel.pth
                s2mel_state_dict.update(self.semantic_codec.s2mel_quantizer_tensors())
                scfg.update(codebook_size=codec_cfg["codebook_size"], codebook_dim=codec_cfg["codebook_dim"], semantic_dim=codec_cfg["hidden_size"])
            self.s2mel = S2Mel(s2mel_state_dict, scfg, device=self.device)
        else:
            missing.append(cfg.get("s2mel_checkpoint", "s2mel.pth"))
        if w2v_bert is not None and not isinstance(w2v_bert, PR.W2vBert):
            st = w2v_stats if w2v_stats is not None else {"mean": torch.zeros(1), "var": torch.ones(1)}
            w2v_bert = PR.W2vBert(w2v_bert, st["mean"], torch.sqrt(st["var"]), device=self.device)
        if w2v_bert is None and have("w2v-bert-2.0") and cfg.get("w2v_stat") and have(cfg["w2v_stat"]):
            w2v_bert = PR.W2vBert.from_dir(os.path.join(model_dir, "w2v-bert-2.0"), os.path.join(model_dir, cfg["w2v_stat"]), self.device)
        self.w2v_bert = w2v_bert
        if w2v_bert is None:
            missing.append("w2v-bert-2.0/ + " + str(cfg.get("w2v_stat", "wav2vec2bert_stats.pt")))
        self.campplus = PR.CamPlus(campplus_state_dict, self.device) if campplus_state_dict is not None else None
        if self.campplus is None:
            missing.append("campplus_cn_common.bin")
        self.prompt = None
        if not missing:
            sp = ((cfg.get("s2mel") or {}).get("preprocess_params") or {})
            spect = sp.get("spect_params") or {}
            mel_args = dict(n_fft=spect.get("n_fft", 1024), win_size=spect.get("win_length", 1024), hop_size=spect.get("hop_length", 256),
                            num_mels=spect.get("n_mels", 80), sampling_rate=sp.get("sr", 22050), fmin=spect.get("fmin", 0),
                            fmax=None if spect.get("fmax", "None") == "None" else 8000)
            self.prompt = PR.PromptEncoder(self.w2v_bert, self.semantic_codec, self.campplus, self.s2mel, self.device, mel_args)
        self.missing_glue = missing
        # emotion matrices (infer_v2.py:168-176): rows grouped per emotion by emo_num
        self.emo_num = list(emo_num if emo_num is not None else cfg.get("emo_num", []))
        for name, given, key in (("emo_matrix", emo_matrix, "emo_matrix"), ("spk_matrix", spk_matrix, "spk_matrix")):
            m = given
            setattr(self, name, torch.split(m.to(self.device), self.emo_num) if m is not None and self.emo_num else None)
        # text front-end (row N4; infer_v2.py:161-165): a given tokenizer, else model_dir/<dataset.bpe_model> with the
        # reference's normaliser (needs WeText, as there), else `glue.tokenize`
        self.tokenizer = tokenizer
        bpe = os.path.join(model_dir, (cfg.get("dataset") or {}).get("bpe_model", "")) if cfg.get("dataset") else None
        if self.tokenizer is None and bpe and os.path.isfile(bpe):
            from .front import TextNormalizer, TextTokenizer

            self.normalizer = TextNormalizer()
            try:
                self.normalizer.load()  # infer_v2.py:162-163: the WeText verbalisers, imported as the reference imports them
                self.tokenizer = TextTokenizer(bpe, self.normalizer)
                logger.info(f"bpe model loaded from: {bpe}")
            except ImportError as e:
                self.missing_glue.append(f"text normalizer (WeTextProcessing not importable: {e})")
        # prompt caches (infer_v2.py:190-197)
        self.cache_spk_audio_prompt = None
        self.cache_spk = None
        self.cache_emo_audio_prompt = None
        self.cache_emo_cond = None

    # ------------------------------------------------------------------ weights: files on rank 0, RCCL everywhere else
    @staticmethod
    def _read_model_dir(files, cfg, model_dir, bcfg, derive_bcfg):
        """Whatever was not handed in as a tensor dict is read from model_dir (INTEGRATION.md layout); tensors only
        (`weights_only=True` / safetensors).  Returns (files, bigvgan cfg)."""
        have = lambda *parts: os.path.exists(os.path.join(model_dir, *parts))
        if files["gpt"] is None and cfg.get("gpt_checkpoint") and os.path.isfile(os.path.join(model_dir, cfg["gpt_checkpoint"])):
            files["gpt"] = load_gpt_checkpoint(os.path.join(model_dir, cfg["gpt_checkpoint"]))
        if files["bigvgan"] is None:
            # `BigVGAN.from_pretrained(cfg.vocoder.name)` (infer_v2.py:154-158, bigvgan.py:436-479) resolves a LOCAL directory holding
            # config.json + bigvgan_generator.pt (or the checkpoint file itself); a hub name is never fetched
            voc = str((cfg.get("vocoder") or {}).get("name", ""))
            cands = [os.path.join(model_dir, voc, "bigvgan_generator.pt"), os.path.join(voc, "bigvgan_generator.pt"), os.path.join(model_dir, voc),
                     os.path.join(model_dir, "bigvgan_generator.pt")]
            for p in cands:
                if voc or p == cands[-1]:
                    if os.path.isfile(p):
                        files["bigvgan"] = load_bigvgan_checkpoint(p)
                        cj = os.path.join(os.path.dirname(p), "config.json")
                        if derive_bcfg and os.path.isfile(cj):
                            import json

                            h = json.load(open(cj))
                            tup = lambda v: tuple(tup(x) for x in v) if isinstance(v, list) else v
                            bcfg.update({k: tup(v) for k, v in h.items() if k in bcfg})
                        break
        if files["codec"] is None and have("semantic_codec", "model.safetensors"):
            from safetensors.torch import load_file

            files["codec"] = load_file(os.path.join(model_dir, "semantic_codec", "model.safetensors"))
        if files["s2mel"] is None and cfg.get("s2mel_checkpoint") and have(cfg["s2mel_checkpoint"]):
            files["s2mel"] = load_s2mel_checkpoint(os.path.join(model_dir, cfg["s2mel_checkpoint"]))
        if files["campplus"] is None and have("campplus_cn_common.bin"):
            files["campplus"] = torch.load(os.path.join(model_dir, "campplus_cn_common.bin"), map_location="cpu", weights_only=True)
        for key in ("emo_matrix", "spk_matrix"):
            if files[key] is None and cfg.get(key) and have(cfg[key]):
                files[key] = torch.load(os.path.join(model_dir, cfg[key]), map_location="cpu", weights_only=True)
        return files, bcfg

    def _exchange_glue(self, files, bcfg, peer):
        """One packed RCCL message with every tensor that is not part of the two device arenas: the non-trunk tensors of gpt.pth
        (text tables, speed embedding, conditioning encoders), s2mel, the semantic codec, CAM++, the emotion matrices -- plus the
        small facts a receiver cannot derive without the files (BigVGAN's config, which dicts exist)."""
        from . import sharding
        from .gpt_engine import GPT_TENSOR_PREFIXES

        pack, meta = None, None
        if not peer:
            pack = {}
            for name in ("gpt", "bigvgan", "s2mel", "codec", "campplus"):
                sd = files[name]
                if sd is None:
                    continue
                for k, v in sd.items():
                    if name == "gpt" and k.startswith(GPT_TENSOR_PREFIXES):
                        continue  # the trunk travels as the packed device arena
                    if name == "bigvgan" and k != "conv_pre.weight":
                        continue  # (only its width is needed by the receiver's constructor; the weights travel as the arena)
                    if torch.is_tensor(v) and v.is_floating_point():
                        pack[f"{name}/{k}"] = v
            for name in ("emo_matrix", "spk_matrix"):
                if files[name] is not None:
                    pack[name] = files[name]
            meta = dict(bcfg=bcfg, present=[k for k, v in files.items() if v is not None])
        recv, meta = sharding.broadcast_state_dict(pack, self.device, src=0, meta=meta)
        if peer:
            out = {k: None for k in files}
            for name in meta["present"]:
                out[name] = {} if name in ("gpt", "bigvgan", "s2mel", "codec", "campplus") else None
            for k, v in recv.items():
                if "/" in k:
                    name, key = k.split("/", 1)
                    out[name][key] = v
                else:
                    out[k] = v
            return out, meta["bcfg"]
        return files, bcfg

    # ------------------------------------------------------------------ helpers mirrored from the reference
    def interval_silence(self, wavs, sampling_rate=22050, interval_silence=200):
        if not wavs or interval_silence <= 0:
            return wavs
        return torch.zeros(wavs[0].size(0), int(sampling_rate * interval_silence / 1000.0))

    def insert_interval_silence(self, wavs, sampling_rate=22050, interval_silence=200):
        if not wavs or interval_silence <= 0:
            return wavs
        sil = torch.zeros(wavs[0].size(0), int(sampling_rate * interval_silence / 1000.0))
        out = []
        for i, w in enumerate(wavs):
            out.append(w)
            if i < len(wavs) - 1:
                out.append(sil)
        return out

    def _prepare_gpt_inputs(self, conds_latent, text_ids):
        """-> fake ids [1, P], embeds [1, P-1, D], mask [1, P] (model_v2.py:598-661; one implementation: pipeline.prepare_gpt_inputs)."""
        embeds, pad, P = prepare_gpt_inputs(self.gpt_cfg, self.text_embedding, self.text_pos_embedding, conds_latent, text_ids)
        mask = torch.ones(1, P, dtype=torch.long, device=self.device)
        mask[0, :pad] = 0
        fake = torch.ones(1, P, dtype=torch.long, device=self.device)
        fake[0, -1] = self.gpt_cfg["start_mel_token"]
        return fake, embeds.unsqueeze(0), mask

    # ---- the once-per-prompt stages: an injected glue method wins, else the built-in prompt encoder, else a clear error
    def _stage(self, name, builtin):
        fn = getattr(self.glue, name, None) if self.glue is not None else None
        if fn is not None and getattr(type(self.glue), name, None) is not getattr(Glue, name, None):
            return fn
        if builtin is None:
            raise NotImplementedError(
                f"IndexTTS2.infer: stage '{name}' has no weights (missing under model_dir: {', '.join(self.missing_glue) or 'emotion matrices'}) "
                f"and no glue= override was given; nothing is downloaded or faked (INTEGRATION.md lists the model_dir layout)")
        return builtin

    def ready(self):
        """True when infer() can synthesise from audio: every stage has weights or an injected override."""
        try:
            self._stage("speaker", self.prompt.speaker if self.prompt else None)
            self._stage("emotion", self.prompt.emotion if self.prompt else None)
            if self.tokenizer is None:
                self._stage("tokenize", None)
            if self.s2mel is None:
                self._stage("s2mel", None)
            if self.cond is None:
                self._stage("merge_emovec", None)
        except NotImplementedError:
            return False
        return True

    def warm_up(self, seconds=3.0, text="Hello, hello. One, two, three."):
        """One synthetic request through EVERY stage (prompt side included) before the worker reports healthy: whatever a fresh
        process pays once -- the BLAS library's code objects for the shapes of this model, the 8-step decode graphs, the
        allocator's pools -- is paid here, at load, not by the first caller (the driver's fresh box spent 182 s in the first
        request of r02; the reference's lifespan loads the model and serves the first request cold, server.py:28-77).  A tone +
        noise prompt of `seconds` s at 22.05 kHz, a short text, at most 48 codes; the prompt caches are cleared afterwards.
        Returns the wall time, or None when the model cannot synthesise (missing stages) or the attempt failed."""
        import io as _io

        if not self.ready():
            return None
        t0 = time.perf_counter()
        rng = np.random.RandomState(0)
        t = np.arange(int(seconds * 22050)) / 22050.0
        x = 0.3 * np.sin(2 * np.pi * 140 * t) + 0.15 * np.sin(2 * np.pi * 280 * t + 1) + 0.02 * rng.randn(t.size)
        buf = _io.BytesIO()
        with wave.open(buf, "wb") as f:
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(22050)
            f.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self.infer(buf.getvalue(), text, None, max_mel_tokens=48)
        except Exception as e:  # a warm-up must never keep a worker from starting
            logger.warning(f"warm-up request failed: {e}")
            return None
        finally:
            self.cache_spk_audio_prompt = self.cache_spk = self.cache_emo_audio_prompt = self.cache_emo_cond = None
        dt = time.perf_counter() - t0
        logger.info(f"warm-up request: {dt:.2f} seconds")
        return dt

    def _builtin_emo_mix(self, emo_vector, style, use_random):
        from .prompt import emo_vector_mix

        if self.emo_matrix is None or self.spk_matrix is None:
            return None
        return emo_vector_mix(emo_vector, style, self.emo_matrix, self.spk_matrix, self.emo_num, use_random)

    # ------------------------------------------------------------------ API
    def infer(self, spk_audio_prompt, text, output_path, emo_audio_prompt=None, emo_alpha=1.0, emo_vector=None,
              use_emo_text=False, emo_text=None, use_random=False, interval_silence=200, verbose=False,
              max_text_tokens_per_segment=120, stream_return=False, more_segment_before=0, **generation_kwargs):
        gen = self.infer_generator(spk_audio_prompt, text, output_path, emo_audio_prompt, emo_alpha, emo_vector, use_emo_text,
                                   emo_text, use_random, interval_silence, verbose, max_text_tokens_per_segment, stream_return,
                                   more_segment_before, **generation_kwargs)
        if stream_return:
            return gen
        try:
            return list(gen)[0]
        except IndexError:
            return None

    # ------------------------------------------------------------------ row N3 at the API level: requests decoded together
    def _many_engine(self, slots):
        """A second decode engine whose slots are shared by the segments of several requests (wide MFMA GEMVs above 4 slots,
        bf16 only; an fp32 model keeps 4)."""
        from . import _lib

        slots = max(1, min(int(slots), _lib.max_batch() if self.use_fp16 else 4))
        if slots not in self._engines:
            # a second engine SHAPE over the same device weights (ixtts_gpt_share_arena): it costs its KV cache only
            eng = GptEngine(self.gpt_cfg, dtype="bf16" if self.use_fp16 else "f32", max_seq=self.gpt.max_seq, max_batch=slots, device=self.device)
            self._engines[slots] = eng.share_arena(self.gpt)
        return self._engines[slots]

    def _beam_group_engine(self, num_beams, n_segments):
        """The engine several beam groups step together on: wide (5..16 slots on the bf16 matrix cores) when the model runs in
        bf16 and there is more than one segment to decode, else None (one group at a time on the register engine, as the
        reference decodes: segment after segment).  `IXTTS_BEAM_GROUPS` caps the groups (0 / 1: off)."""
        from . import _lib

        cap = int(os.environ.get("IXTTS_BEAM_GROUPS", "5"))
        groups = min(cap, _lib.max_batch() // num_beams, n_segments)
        if not self.use_fp16 or groups < 2 or groups * num_beams <= 4:
            return None
        want = min(cap, _lib.max_batch() // num_beams) * num_beams  # one engine shape per worker, whatever the request's segment count
        eng = self._many_engine(want)
        return eng if eng.max_batch >= 2 * num_beams else None

    def _check_codes(self, codes):
        """The reference would index the semantic codec's codebook out of range with such a code (a device-side assert on a
        GPU, which takes the worker's context with it): refuse on the host instead."""
        if self.s2mel is not None and codes.numel():
            rows = self.s2mel.W["quantizer.codebook.weight"].shape[0]
            top = int(codes.max())
            if top >= rows:
                raise ValueError(f"mel code {top} outside the semantic codec's codebook ({rows} entries)")

    @torch.no_grad()
    def infer_many(self, requests, interval_silence=200, max_text_tokens_per_segment=120, decode_slots=8, **generation_kwargs):
        """Several `/tts` requests served TOGETHER (the worker's global lock, server.py:25,384, replaced by the
        decode scheduler): every request's segments share the decode slots -- the weights are read once per step for all of them
        -- and the post-decode stages run per segment as in `infer`.  Each request is a dict with `spk_audio_prompt`, `text` and
        optionally `emo_audio_prompt`, `emo_alpha`, `emo_vector`, `use_random`.  Generation kwargs and defaults are `infer`'s:
        with `num_beams > 1` (the served default, 3) every segment is a beam GROUP and floor(decode_slots / num_beams) groups step
        together (bf16 engines; an fp32 model decodes group after group); `num_beams=1` samples without beams, one slot per
        segment (argmax when `top_k == 1`).  Returns one entry per request: `(22050, int16 [N, 1])`, None (empty text), or the
        EXCEPTION that request raised (bad prompt audio, a code outside the codebook ...) -- one request's failure leaves the
        others of the batch alone."""
        from . import _lib
        from .scheduler import BeamGroupScheduler, DecodeScheduler, Segment

        generation_kwargs.pop("do_sample", True)
        top_p = generation_kwargs.pop("top_p", 0.8)
        top_k = generation_kwargs.pop("top_k", 30)
        temperature = generation_kwargs.pop("temperature", 0.8)
        num_beams = int(generation_kwargs.pop("num_beams", 3))
        length_penalty = generation_kwargs.pop("length_penalty", 0.0)
        repetition_penalty = generation_kwargs.pop("repetition_penalty", 10.0)
        max_mel_tokens = generation_kwargs.pop("max_mel_tokens", 1500)
        typical_mass = float(generation_kwargs.get("typical_mass", 0.9)) if generation_kwargs.get("typical_sampling") else 0.0
        speaker_fn = self._stage("speaker", self.prompt.speaker if self.prompt else None)
        emotion_fn = self._stage("emotion", self.prompt.emotion if self.prompt else None)
        if num_beams > 1:
            if not (2 <= num_beams <= 4 and 1 <= top_k <= 128):
                raise NotImplementedError("beam-sample on the device: 2 <= num_beams <= 4, 1 <= top_k <= 128")
            groups = max(1, min(int(decode_slots), _lib.max_batch()) // num_beams)
            eng = self._many_engine(groups * num_beams) if self.use_fp16 and groups * num_beams > 4 else self.gpt
        else:
            eng = self._many_engine(decode_slots)
        start = time.perf_counter()
        plans, todo = [None] * len(requests), []
        failed = {}
        for ri, rq in enumerate(requests):
            try:
                spk_prompt, emo_prompt = rq["spk_audio_prompt"], rq.get("emo_audio_prompt")
                emo_alpha, emo_vector = rq.get("emo_alpha", 1.0), rq.get("emo_vector")
                if emo_vector is not None:  # infer_v2.py:476-505
                    emo_prompt = None
                    scale = max(0.0, min(1.0, emo_alpha))
                    if scale != 1.0:
                        emo_vector = [int(x * scale * 10000) / 10000 for x in emo_vector]
                emo_is_spk = emo_prompt is None  # (then both emotion-encoder passes see the same features: computed once)
                if emo_prompt is None:
                    emo_prompt, emo_alpha = spk_prompt, 1.0
                if self.cache_spk is None or not _same_prompt(self.cache_spk_audio_prompt, spk_prompt):
                    self.cache_spk, self.cache_spk_audio_prompt = None, None  # (a failing encode must not leave a stale pair behind)
                    self.cache_spk, self.cache_spk_audio_prompt = speaker_fn(spk_prompt), spk_prompt
                spk = self.cache_spk
                if self.cache_emo_cond is None or not _same_prompt(self.cache_emo_audio_prompt, emo_prompt):
                    self.cache_emo_cond, self.cache_emo_audio_prompt = None, None
                    self.cache_emo_cond, self.cache_emo_audio_prompt = emotion_fn(emo_prompt), emo_prompt
                emo_cond = self.cache_emo_cond
                if self.cond is not None:
                    cond32, emovec = self.cond.encode_prompt(spk["spk_cond_emb"], None if emo_is_spk else emo_cond, emo_alpha)
                else:
                    emovec = self._stage("merge_emovec", None)(spk["spk_cond_emb"], emo_cond, emo_alpha)
                    cond32 = self._stage("get_conditioning", None)(spk["spk_cond_emb"])
                if emo_vector is not None:
                    mix = self._stage("emo_vector_mix", (lambda v, st, r: self._builtin_emo_mix(v, st, r)) if self.emo_matrix is not None else None)
                    emovec_mat, weight_sum = mix(emo_vector, spk["style"], rq.get("use_random", False))
                    emovec = emovec_mat + (1 - weight_sum) * emovec
                cl = torch.cat((cond32 + emovec.reshape(1, -1), self.speed_emb[1:2], self.speed_emb[0:1]), 0)
                if self.tokenizer is not None:
                    toks = self.tokenizer.tokenize(rq["text"])
                    segments = [self.tokenizer.convert_tokens_to_ids(sent) for sent in self.tokenizer.split_segments(toks, max_text_tokens_per_segment)]
                else:
                    segments = self._stage("tokenize", None)(rq["text"], max_text_tokens_per_segment, 0)
                mine = []
                for si, ids in enumerate(segments):
                    tt = torch.as_tensor(ids, dtype=torch.int32, device=self.device).reshape(-1)
                    fake, embeds, mask = self._prepare_gpt_inputs(cl, tt)
                    n_pad = int((mask == 0).sum().item())
                    max_new = max(0, min(max_mel_tokens, eng.max_seq - fake.shape[1] - 2, self.gpt_cfg["max_mel_tokens"] - 1))
                    mine.append(Segment(ri, si, embeds[0], n_pad, max_new))
                plans[ri] = dict(spk=spk, cl=cl, segments=segments, codes=[None] * len(segments))
                todo += mine  # only once the whole request is known to be well-formed
            except Exception as e:  # this request only
                failed[ri] = e
        sampler = dict(repetition_penalty=repetition_penalty, temperature=temperature, top_k=top_k, top_p=top_p, seed=int(generation_kwargs.get("seed", 0)),
                       typical_mass=typical_mass)
        if todo and num_beams > 1:
            BeamGroupScheduler(eng, num_beams).run(todo, lambda seg, ids, score: plans[seg.request]["codes"].__setitem__(seg.index, ids),
                                                   length_penalty=length_penalty, **sampler)
        elif todo:
            DecodeScheduler(eng, eng.max_batch, self.stop_mel_token).run(
                todo, lambda seg, ids: plans[seg.request]["codes"].__setitem__(seg.index, ids), do_sample=top_k != 1, **sampler)
        torch.cuda.synchronize(self.device)
        t_decode = time.perf_counter() - start
        out = []
        for ri, plan in enumerate(plans):
            if ri in failed:
                out.append(failed[ri])
                continue
            try:
                wavs = []
                for ids, seg in zip(plan["codes"], plan["segments"]):
                    row = torch.from_numpy(np.asarray(ids).astype(np.int64)).to(self.device)
                    stops = (row == self.stop_mel_token).nonzero(as_tuple=False)
                    n = int(stops[0]) if stops.numel() else row.numel()
                    if n == 0:
                        continue
                    codes = row[:n].reshape(1, -1)
                    self._check_codes(codes)
                    tt = torch.as_tensor(seg, dtype=torch.int32, device=self.device).reshape(-1)
                    t = torch.cat((tt.new_tensor([self.gpt_cfg["start_text_token"]]), tt, tt.new_tensor([self.gpt_cfg["stop_text_token"]]))).long()
                    prefix = torch.cat((plan["cl"], self.text_embedding[t] + self.text_pos_embedding[: t.numel()]), 0)
                    latent = self.gpt.latent(prefix, codes[0]).unsqueeze(0)
                    lens = torch.tensor([n], dtype=torch.long, device=self.device)
                    spk = plan["spk"]
                    if self.s2mel is not None:
                        mel = self.s2mel(latent, codes, lens, spk["prompt_condition"], spk["ref_mel"], spk["style"], n_timesteps=25, inference_cfg_rate=0.7)
                    else:
                        mel = self._stage("s2mel", None)(latent, codes, lens, spk)
                    wav = torch.clamp(32767 * self.bigvgan(mel.float()).squeeze().unsqueeze(0), -32767.0, 32767.0)
                    wavs.append(wav.cpu())
                if not wavs:
                    out.append(None)
                    continue
                wav = torch.cat(self.insert_interval_silence(wavs, sampling_rate=22050, interval_silence=interval_silence), dim=1)
                out.append((22050, wav.type(torch.int16).numpy().T))
            except Exception as e:  # this request only
                out.append(e)
        total = time.perf_counter() - start
        audio = sum(o[1].shape[0] for o in out if isinstance(o, tuple)) / 22050.0
        logger.info(f"infer_many: {len(requests)} requests ({sum(isinstance(o, Exception) for o in out)} failed), {len(todo)} segments, "
                    f"decode {t_decode:.2f} s, total {total:.2f} s, audio {audio:.2f} s, RTF {total / max(audio, 1e-9):.4f}")
        self.last_timing = dict(gpt_gen_time=t_decode, total=total, audio_length=audio)
        return out

    def infer_generator(self, spk_audio_prompt, text, output_path, emo_audio_prompt=None, emo_alpha=1.0, emo_vector=None,
                        use_emo_text=False, emo_text=None, use_random=False, interval_silence=200, verbose=False,
                        max_text_tokens_per_segment=120, stream_return=False, quick_streaming_tokens=0, **generation_kwargs):
        logger.info("Starting inference...")
        start_time = time.perf_counter()
        glue = self.glue
        speaker_fn = self._stage("speaker", self.prompt.speaker if self.prompt else None)
        emotion_fn = self._stage("emotion", self.prompt.emotion if self.prompt else None)
        if use_emo_text:
            raise NotImplementedError("use_emo_text needs the Qwen emotion model (infer_v2.py:481-488), out of scope")
        if emo_vector is not None:
            emo_audio_prompt = None
            scale = max(0.0, min(1.0, emo_alpha))
            if scale != 1.0:
                emo_vector = [int(x * scale * 10000) / 10000 for x in emo_vector]
        if emo_audio_prompt is None:
            emo_audio_prompt = spk_audio_prompt
            emo_alpha = 1.0
        if self.cache_spk is None or not _same_prompt(self.cache_spk_audio_prompt, spk_audio_prompt):
            self.cache_spk = speaker_fn(spk_audio_prompt)
            self.cache_spk_audio_prompt = spk_audio_prompt
        spk = self.cache_spk
        emovec_mat = weight_sum = None
        if emo_vector is not None:
            mix = self._stage("emo_vector_mix", (lambda v, st, r: self._builtin_emo_mix(v, st, r)) if self.emo_matrix is not None else None)
            emovec_mat, weight_sum = mix(emo_vector, spk["style"], use_random)
        if self.cache_emo_cond is None or not _same_prompt(self.cache_emo_audio_prompt, emo_audio_prompt):
            self.cache_emo_cond = emotion_fn(emo_audio_prompt)
            self.cache_emo_audio_prompt = emo_audio_prompt
        emo_cond_emb = self.cache_emo_cond

        if self.tokenizer is not None:  # infer_v2.py:582-589,617
            text_tokens_list = self.tokenizer.tokenize(text)
            text_token_ids = self.tokenizer.convert_tokens_to_ids(text_tokens_list)
            unk = self.tokenizer.unk_token_id
            if unk in text_token_ids:
                logger.warning(f"Input text contains {text_token_ids.count(unk)} unknown tokens (id={unk})")
                logger.warning(f"Tokens which can't be encoded: {[t for t, i in zip(text_tokens_list, text_token_ids) if i == unk]}")
            segments = [self.tokenizer.convert_tokens_to_ids(sent) for sent in
                        self.tokenizer.split_segments(text_tokens_list, max_text_tokens_per_segment, quick_streaming_tokens=quick_streaming_tokens)]
        else:
            segments = self._stage("tokenize", None)(text, max_text_tokens_per_segment, quick_streaming_tokens)
        # generation kwargs and their defaults (infer_v2.py:598-606); do_sample is popped and then forced True (:648)
        generation_kwargs.pop("do_sample", True)
        top_p = generation_kwargs.pop("top_p", 0.8)
        top_k = generation_kwargs.pop("top_k", 30)
        temperature = generation_kwargs.pop("temperature", 0.8)
        length_penalty = generation_kwargs.pop("length_penalty", 0.0)
        num_beams = generation_kwargs.pop("num_beams", 3)
        repetition_penalty = generation_kwargs.pop("repetition_penalty", 10.0)
        max_mel_tokens = generation_kwargs.pop("max_mel_tokens", 1500)
        sampling_rate = 22050

        wavs = []
        gpt_gen_time = gpt_forward_time = s2mel_time = bigvgan_time = 0.0
        has_warned = False
        silence = None
        req_emovec = req_cond32 = None
        if self.cond is not None:
            # merge_emovec + get_conditioning are functions of the request's prompts only; the reference recomputes them per
            # segment (infer_v2.py:629-635, model_v2.py:684-689) -- computed once here, same values.  The length arguments
            # are the reference's: spk_cond_emb.shape[-1] (= 1024, the feature size: no frame is ever masked).
            m0 = time.perf_counter()
            sc = spk["spk_cond_emb"].to(self.device, torch.float32)
            ec = emo_cond_emb.to(self.device, torch.float32)
            ls, le = torch.tensor([sc.shape[-1]], device=self.device), torch.tensor([ec.shape[-1]], device=self.device)
            req_emovec = self.cond.merge_emovec(sc, ec, ls, le, alpha=emo_alpha)
            req_cond32 = self.cond.get_conditioning(sc.transpose(1, 2), ls)[0]
            gpt_gen_time += time.perf_counter() - m0
        def conds_for_segment():
            emovec = req_emovec if req_emovec is not None else self._stage("merge_emovec", None)(spk["spk_cond_emb"], emo_cond_emb, emo_alpha)
            if emo_vector is not None:
                emovec = emovec_mat + (1 - weight_sum) * emovec
            cond32 = req_cond32 if req_cond32 is not None else self._stage("get_conditioning", None)(spk["spk_cond_emb"])
            # inference_speech (model_v2.py:693-734)
            return torch.cat((cond32 + emovec.reshape(1, -1), self.speed_emb[1:2], self.speed_emb[0:1]), 0)

        # Without beams the segments are independent sequences: decode them together, the weights are read once per step for
        # all of them (the reference decodes segment after segment, infer_v2.py:616; tokens per segment are the same for
        # greedy; with sampling each slot draws from its own counter-based stream).  Beam search keeps one segment at a time.
        pre = None
        beam_eng = None
        if num_beams > 1 and len(segments) > 1 and not stream_return and not generation_kwargs.get("logits_processor") and 1 <= top_k <= 128:
            beam_eng = self._beam_group_engine(num_beams, len(segments))
        if beam_eng is not None:
            # The served default (num_beams=3): every segment is a beam group of num_beams slots and the groups step TOGETHER on the
            # wide engine -- the weights are read once per step for all of them (the reference runs `inference_speech` segment after
            # segment, infer_v2.py:616-658).  Each segment has its own scorer state and its own random stream (segment index).
            from .scheduler import BeamGroupScheduler, Segment

            m0 = time.perf_counter()
            todo = []
            for i, sent_ids in enumerate(segments):
                tt = torch.as_tensor(sent_ids, dtype=torch.int32, device=self.device).reshape(-1)
                cl = conds_for_segment()
                fake, embeds, mask = self._prepare_gpt_inputs(cl, tt)
                n_pad = int((mask == 0).sum().item())
                max_new = max(0, min(max_mel_tokens, beam_eng.max_seq - fake.shape[1] - 2, self.gpt_cfg["max_mel_tokens"] - 1))
                todo.append(Segment(0, i, embeds[0], n_pad, max_new, payload=cl, stream=i))
            pre = [None] * len(segments)
            BeamGroupScheduler(beam_eng, num_beams).run(
                todo, lambda seg, ids, score: pre.__setitem__(seg.index, (ids, seg.payload)), repetition_penalty=repetition_penalty,
                temperature=temperature, top_k=top_k, top_p=top_p, seed=int(generation_kwargs.get("seed", 0)), length_penalty=length_penalty,
                typical_mass=float(generation_kwargs.get("typical_mass", 0.9)) if generation_kwargs.get("typical_sampling") else 0.0)
            torch.cuda.synchronize(self.device)
            gpt_gen_time += time.perf_counter() - m0
        elif num_beams == 1 and len(segments) > 1 and not stream_return and not generation_kwargs.get("logits_processor"):
            from .scheduler import DecodeScheduler, Segment

            m0 = time.perf_counter()
            greedy = top_k == 1
            todo = []
            for i, sent_ids in enumerate(segments):
                tt = torch.as_tensor(sent_ids, dtype=torch.int32, device=self.device).reshape(-1)
                cl = conds_for_segment()
                fake, embeds, mask = self._prepare_gpt_inputs(cl, tt)
                n_pad = int((mask == 0).sum().item())
                max_new = max(0, min(max_mel_tokens, self.gpt.max_seq - fake.shape[1] - 2, self.gpt_cfg["max_mel_tokens"] - 1))
                todo.append(Segment(0, i, embeds[0], n_pad, max_new, payload=cl))
            pre = [None] * len(segments)
            DecodeScheduler(self.gpt, self.gpt.max_batch, self.stop_mel_token).run(
                todo, lambda seg, ids: pre.__setitem__(seg.index, (ids, seg.payload)), repetition_penalty=repetition_penalty,
                temperature=temperature, top_k=top_k, top_p=top_p, do_sample=not greedy, seed=int(generation_kwargs.get("seed", 0)),
                typical_mass=float(generation_kwargs.get("typical_mass", 0.9)) if generation_kwargs.get("typical_sampling") else 0.0)
            torch.cuda.synchronize(self.device)
            gpt_gen_time += time.perf_counter() - m0
        for seg_index, sent_ids in enumerate(segments):
            text_tokens = torch.as_tensor(sent_ids, dtype=torch.int32, device=self.device).reshape(-1)
            m0 = time.perf_counter()
            if pre is not None:
                ids, conds_latent = pre[seg_index]
                codes = torch.from_numpy(np.asarray(ids).astype(np.int64)).reshape(1, -1).to(self.device)
            else:
                conds_latent = conds_for_segment()
                fake, embeds, mask = self._prepare_gpt_inputs(conds_latent, text_tokens)
                self.gpt.store_mel_emb(embeds)
                trunc = fake.shape[1]
                out = self.gpt.generate(fake, bos_token_id=self.gpt_cfg["start_mel_token"], pad_token_id=self.stop_mel_token,
                                        eos_token_id=self.stop_mel_token, attention_mask=mask, max_length=trunc + max_mel_tokens,
                                        num_return_sequences=1, do_sample=True, top_p=top_p, top_k=top_k, temperature=temperature,
                                        num_beams=num_beams, repetition_penalty=repetition_penalty, length_penalty=length_penalty,
                                        **generation_kwargs)
                codes = out[:, trunc:]
                torch.cuda.synchronize(self.device)
                gpt_gen_time += time.perf_counter() - m0
            if codes.shape[1] == 0:  # no room left to generate (max_mel_tokens == 0 or a prompt as long as max_seq)
                raise RuntimeError(f"no mel codes could be generated for segment {seg_index} (max_mel_tokens={max_mel_tokens}, max_seq={self.gpt.max_seq})")
            if not has_warned and bool((codes[:, -1] != self.stop_mel_token).any()):
                warnings.warn(f"WARN: generation stopped due to exceeding `max_mel_tokens` ({max_mel_tokens}). "
                              f"Input text tokens: {text_tokens.shape[0]}. Consider reducing `max_text_tokens_per_segment`"
                              f"({max_text_tokens_per_segment}) or increasing `max_mel_tokens`.", category=RuntimeWarning)
                has_warned = True
            # trim at the first stop token (infer_v2.py:676-687)
            row = codes[0]
            stops = (row == self.stop_mel_token).nonzero(as_tuple=False)
            code_len = int(stops[0]) if stops.numel() else row.numel()
            codes = codes[:, :code_len]
            code_lens = torch.tensor([code_len], dtype=torch.long, device=self.device)

            m0 = time.perf_counter()
            t = torch.cat((text_tokens.new_tensor([self.gpt_cfg["start_text_token"]]), text_tokens,
                           text_tokens.new_tensor([self.gpt_cfg["stop_text_token"]]))).long()
            prefix = torch.cat((conds_latent, self.text_embedding[t] + self.text_pos_embedding[: t.numel()]), 0)
            latent = self.gpt.latent(prefix, codes[0]).unsqueeze(0)  # UnifiedVoice.forward (model_v2.py:554-596)
            torch.cuda.synchronize(self.device)
            gpt_forward_time += time.perf_counter() - m0

            m0 = time.perf_counter()
            self._check_codes(codes)
            if self.s2mel is not None:  # infer_v2.py:713-731
                mel = self.s2mel(latent, codes, code_lens, spk["prompt_condition"], spk["ref_mel"], spk["style"],
                                 n_timesteps=25, inference_cfg_rate=0.7)
            else:
                mel = self._stage("s2mel", None)(latent, codes, code_lens, spk)
            torch.cuda.synchronize(self.device)
            s2mel_time += time.perf_counter() - m0

            m0 = time.perf_counter()
            wav = self.bigvgan(mel.float()).squeeze().unsqueeze(0)
            torch.cuda.synchronize(self.device)
            bigvgan_time += time.perf_counter() - m0
            wav = torch.clamp(32767 * wav, -32767.0, 32767.0)
            wavs.append(wav.cpu())
            if stream_return:
                yield wav.cpu()
                if silence is None:
                    silence = self.interval_silence(wavs, sampling_rate=sampling_rate, interval_silence=interval_silence)
                yield silence
        end_time = time.perf_counter()
        if not wavs:
            return
        wavs = self.insert_interval_silence(wavs, sampling_rate=sampling_rate, interval_silence=interval_silence)
        wav = torch.cat(wavs, dim=1)
        wav_length = wav.shape[-1] / sampling_rate
        logger.info(f"gpt_gen_time: {gpt_gen_time:.2f} seconds")
        logger.info(f"gpt_forward_time: {gpt_forward_time:.2f} seconds")
        logger.info(f"s2mel_time: {s2mel_time:.2f} seconds")
        logger.info(f"bigvgan_time: {bigvgan_time:.2f} seconds")
        logger.info(f"Total inference time: {end_time - start_time:.2f} seconds")
        logger.info(f"Generated audio length: {wav_length:.2f} seconds")
        logger.info(f"RTF: {(end_time - start_time) / wav_length:.4f}")
        self.last_timing = dict(gpt_gen_time=gpt_gen_time, gpt_forward_time=gpt_forward_time, s2mel_time=s2mel_time,
                                bigvgan_time=bigvgan_time, total=end_time - start_time, audio_length=wav_length)
        wav = wav.cpu()
        if output_path:
            if os.path.isfile(output_path):
                os.remove(output_path)
            if os.path.dirname(output_path) != "":
                os.makedirs(os.path.dirname(output_path), exist_ok=True)
            pcm = wav.type(torch.int16).numpy()  # truncation toward zero, as torchaudio.save(wav.type(torch.int16)) (infer_v2.py:772)
            with wave.open(output_path, "wb") as f:
                f.setnchannels(pcm.shape[0])
                f.setsampwidth(2)
                f.setframerate(sampling_rate)
                f.writeframes(pcm.T.astype("<i2").tobytes())
            if stream_return:
                return
            yield output_path
        else:
            if stream_return:
                return
            yield (sampling_rate, wav.type(torch.int16).numpy().T)
