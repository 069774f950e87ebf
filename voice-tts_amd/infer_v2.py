"""Host-side mirror of `indextts.infer_v2.IndexTTS2` for the hot path this repo builds.

Same constructor and `infer(...)` signature, return values, wav format and log lines as the
reference (`indextts/infer_v2.py:36-45,438-461,463-783`).  The two hot stages -- the GPT
(`inference_speech` + latent `forward`) and BigVGAN -- run in libixtts_hip.so.  Of the stages
`north_star` leaves to PyTorch glue this repository builds the conformer/perceiver conditioners
(`conditioning.py`, used when the GPT checkpoint carries their weights) and the s2mel
length-regulator + CFM (`s2mel.py`, used when `s2mel_state_dict` is given); audio loading,
w2v-bert features, the semantic codec, CAM++ and the text front-end are NOT implemented: they are
supplied through a `glue` object (see `Glue`), and `infer()` raises a clear `NotImplementedError`
when it is missing.  That keeps the orchestration (segment loop, generation kwargs, stop-token
trimming, PCM conversion, silence insertion, streaming) testable today and lets the reference's
own modules be plugged in unchanged where they are available.
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
from .weights import BIGVGAN_CFG, GPT_CFG, load_bigvgan_checkpoint, load_gpt_checkpoint

logger = logging.getLogger("indextts.infer_v2")


class Glue:
    """Interface of the PyTorch-hosted stages.  All tensors on `device`."""

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
    def __init__(self, cfg_path="models/IndexTTS/config.yaml", model_dir="models/IndexTTS", use_fp16=False, device=None,
                 use_cuda_kernel=None, use_deepspeed=False, *, glue=None, gpt_state_dict=None, bigvgan_state_dict=None,
                 s2mel_state_dict=None, gpt_cfg=None, bigvgan_cfg=None, cond_cfg=None, s2mel_cfg=None, tokenizer=None,
                 max_seq=2048, max_frames=4096):
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
        self.stop_mel_token = gcfg["stop_mel_token"]
        # reference precision: fp16 GPT under use_fp16 (infer_v2.py:79,88-89); here bf16 is the throughput mode
        self.gpt = GptEngine(gcfg, dtype="bf16" if use_fp16 else "f32", max_seq=max_seq, max_batch=3, device=self.device)
        self.bigvgan = BigVGAN(bcfg, use_cuda_kernel=True, max_frames=max_frames, device=self.device)
        if gpt_state_dict is None and cfg.get("gpt_checkpoint") and os.path.isfile(os.path.join(model_dir, cfg["gpt_checkpoint"])):
            gpt_state_dict = load_gpt_checkpoint(os.path.join(model_dir, cfg["gpt_checkpoint"]))
        if gpt_state_dict is None:
            raise FileNotFoundError("no GPT weights: pass gpt_state_dict=... or provide model_dir/gpt_checkpoint (checkpoint.py:25-34)")
        if bigvgan_state_dict is None:
            p = os.path.join(model_dir, "bigvgan_generator.pt")
            if os.path.isfile(p):
                bigvgan_state_dict = load_bigvgan_checkpoint(p)
        if bigvgan_state_dict is None:
            raise FileNotFoundError("no BigVGAN weights: pass bigvgan_state_dict=... or provide model_dir/bigvgan_generator.pt")
        self.gpt.load_state_dict(gpt_state_dict)
        self.bigvgan.load_state_dict(bigvgan_state_dict)
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

            self.cond = Conditioning(gpt_state_dict, COND_CFG if cond_cfg is None else cond_cfg, device=self.device)
        self.s2mel = None
        if s2mel_state_dict is not None:
            from .s2mel import S2MEL_CFG, S2Mel

            self.s2mel = S2Mel(s2mel_state_dict, S2MEL_CFG if s2mel_cfg is None else s2mel_cfg, device=self.device)
        # text front-end (row N4; infer_v2.py:161-165): a given tokenizer, else model_dir/<dataset.bpe_model> with the
        # reference's normaliser (needs WeText, as there), else `glue.tokenize`
        self.tokenizer = tokenizer
        bpe = os.path.join(model_dir, (cfg.get("dataset") or {}).get("bpe_model", "")) if cfg.get("dataset") else None
        if self.tokenizer is None and bpe and os.path.isfile(bpe):
            from .front import TextNormalizer, TextTokenizer

            self.normalizer = TextNormalizer()
            self.tokenizer = TextTokenizer(bpe, self.normalizer)
            logger.info(f"bpe model loaded from: {bpe}")
        # prompt caches (infer_v2.py:190-197)
        self.cache_spk_audio_prompt = None
        self.cache_spk = None
        self.cache_emo_audio_prompt = None
        self.cache_emo_cond = None

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
        """UnifiedVoice.prepare_gpt_inputs (model_v2.py:598-661): -> fake ids [1,P], embeds [1,P-1,D], mask [1,P]."""
        c = self.gpt_cfg
        t = torch.as_tensor(text_ids, dtype=torch.long, device=self.device).reshape(-1)
        L = t.numel()
        t = t[(t != c["stop_text_token"]) & (t != c["start_text_token"])]
        t = torch.cat((t.new_tensor([c["start_text_token"]]), t, t.new_tensor([c["stop_text_token"]])))
        temb = self.text_embedding[t] + self.text_pos_embedding[: t.numel()]
        pad = L + 2 - t.numel()
        parts = [conds_latent, temb]
        if pad > 0:
            parts.insert(0, torch.zeros(pad, self.model_dim, device=self.device))
        embeds = torch.cat(parts, 0)
        P = embeds.shape[0] + 1
        mask = torch.ones(1, P, dtype=torch.long, device=self.device)
        mask[0, :pad] = 0
        fake = torch.ones(1, P, dtype=torch.long, device=self.device)
        fake[0, -1] = c["start_mel_token"]
        return fake, embeds.unsqueeze(0), mask

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

    def infer_generator(self, spk_audio_prompt, text, output_path, emo_audio_prompt=None, emo_alpha=1.0, emo_vector=None,
                        use_emo_text=False, emo_text=None, use_random=False, interval_silence=200, verbose=False,
                        max_text_tokens_per_segment=120, stream_return=False, quick_streaming_tokens=0, **generation_kwargs):
        if self.glue is None:
            raise NotImplementedError(
                "IndexTTS2.infer needs the PyTorch glue stages (audio features, conditioners, s2mel, text front-end), which this "
                "repository does not build (DESIGN.md section 7); pass glue=<voice_tts_amd.infer_v2.Glue implementation>")
        logger.info("Starting inference...")
        start_time = time.perf_counter()
        glue = self.glue
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
        if self.cache_spk is None or self.cache_spk_audio_prompt is not spk_audio_prompt:
            self.cache_spk = glue.speaker(spk_audio_prompt)
            self.cache_spk_audio_prompt = spk_audio_prompt
        spk = self.cache_spk
        emovec_mat = weight_sum = None
        if emo_vector is not None:
            emovec_mat, weight_sum = glue.emo_vector_mix(emo_vector, spk["style"], use_random)
        if self.cache_emo_cond is None or self.cache_emo_audio_prompt is not emo_audio_prompt:
            self.cache_emo_cond = glue.emotion(emo_audio_prompt)
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
            segments = glue.tokenize(text, max_text_tokens_per_segment, quick_streaming_tokens)
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
            emovec = req_emovec if req_emovec is not None else glue.merge_emovec(spk["spk_cond_emb"], emo_cond_emb, emo_alpha)
            if emo_vector is not None:
                emovec = emovec_mat + (1 - weight_sum) * emovec
            cond32 = req_cond32 if req_cond32 is not None else glue.get_conditioning(spk["spk_cond_emb"])
            # inference_speech (model_v2.py:693-734)
            return torch.cat((cond32 + emovec.reshape(1, -1), self.speed_emb[1:2], self.speed_emb[0:1]), 0)

        # Without beams the segments are independent sequences: decode them together, the weights are read once per step for
        # all of them (the reference decodes segment after segment, infer_v2.py:616; tokens per segment are the same for
        # greedy; with sampling each slot draws from its own counter-based stream).  Beam search keeps one segment at a time.
        pre = None
        if num_beams == 1 and len(segments) > 1 and not stream_return and not generation_kwargs.get("logits_processor"):
            from .scheduler import DecodeScheduler, Segment

            m0 = time.perf_counter()
            greedy = top_k == 1
            todo = []
            for i, sent_ids in enumerate(segments):
                tt = torch.as_tensor(sent_ids, dtype=torch.int32, device=self.device).reshape(-1)
                cl = conds_for_segment()
                fake, embeds, mask = self._prepare_gpt_inputs(cl, tt)
                n_pad = int((mask == 0).sum().item())
                max_new = max(0, min(max_mel_tokens, self.gpt.max_seq - fake.shape[1] - 2))
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
            if self.s2mel is not None:  # infer_v2.py:713-731
                mel = self.s2mel(latent, codes, code_lens, spk["prompt_condition"], spk["ref_mel"], spk["style"],
                                 n_timesteps=25, inference_cfg_rate=0.7)
            else:
                mel = glue.s2mel(latent, codes, code_lens, spk)
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
