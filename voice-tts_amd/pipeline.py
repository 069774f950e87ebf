"""Host-side mirror of the hot loop of `IndexTTS2.infer_generator` (indextts/infer_v2.py:616-749)
for the two stages this repo builds: the autoregressive GPT (`inference_speech` + latent
`forward`) and the BigVGAN vocoder.  PyTorch is plumbing only (device tensors, the tiny
embedding gathers of `prepare_gpt_inputs`); all heavy arithmetic runs in libixtts_hip.so.

The stages between them that `north_star` leaves to PyTorch glue (conditioning encoders,
s2mel length-regulator + CFM) are NOT part of this package; callers hand in
`conds_latent` (model_v2.py:696) and the mel spectrogram (infer_v2.py:731).
"""
import numpy as np
import torch

from .bigvgan import BigVGAN
from .gpt_engine import GptEngine
from .weights import BIGVGAN_CFG, GPT_CFG

SAMPLE_RATE = 22050  # infer_v2.py:607


class _RawDeviceBuffer:
    """Expose a raw device pointer to torch without copying (for the RCCL weight broadcast)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def arena_tensor(ptr, nbytes, device):
    return torch.as_tensor(_RawDeviceBuffer(ptr, nbytes), device=device)


def prepare_gpt_inputs(cfg, text_embedding, text_pos_embedding, conds_latent, text_ids):
    """UnifiedVoice.prepare_gpt_inputs (model_v2.py:598-661), one sequence: start/stop text ids found INSIDE the text are
    dropped and made up for by zero rows on the left (mask 0 there), then [start] text [stop] is embedded.

    conds_latent [34, D] device; text_ids int [L].  Returns (embeds [P-1, D], n_left_pad, P)."""
    device = text_embedding.device
    t = torch.as_tensor(text_ids, dtype=torch.long, device=device).reshape(-1)
    L = t.numel()
    t = t[(t != cfg["stop_text_token"]) & (t != cfg["start_text_token"])]
    t = torch.cat((t.new_tensor([cfg["start_text_token"]]), t, t.new_tensor([cfg["stop_text_token"]])))
    temb = text_embedding[t] + text_pos_embedding[: t.numel()]
    pad = L + 2 - t.numel()
    parts = [conds_latent.to(device, torch.float32), temb]
    if pad > 0:
        parts.insert(0, torch.zeros(pad, temb.shape[1], device=device))
    embeds = torch.cat(parts, 0)
    return embeds, pad, embeds.shape[0] + 1


class HotPath:
    def __init__(self, gpt_cfg=None, bigvgan_cfg=None, dtype="bf16", device="cuda:0", max_batch=2, max_seq=2048,
                 max_frames=2048, fast_sin=False):
        self.device = torch.device(device)
        self.gpt_cfg = dict(GPT_CFG if gpt_cfg is None else gpt_cfg)
        self.bigvgan_cfg = dict(BIGVGAN_CFG if bigvgan_cfg is None else bigvgan_cfg)
        self.gpt = GptEngine(self.gpt_cfg, dtype=dtype, max_seq=max_seq, max_batch=max_batch, device=self.device)
        self.bigvgan = BigVGAN(self.bigvgan_cfg, max_frames=max_frames, fast_sin=fast_sin, device=self.device)
        D = self.gpt_cfg["model_dim"]
        # glue tables for prepare_gpt_inputs / the latent prefix (model_v2.py:380,388-390,402)
        self.text_embedding = torch.zeros(self.gpt_cfg["number_text_tokens"] + 1, D, device=self.device)
        self.text_pos_embedding = torch.zeros(self.gpt_cfg["max_text_tokens"] + 2, D, device=self.device)
        self.speed_emb = torch.zeros(2, D, device=self.device)

    # ------------------------------------------------------------------ weights
    def load(self, gpt_sd, bigvgan_sd):
        self.gpt.load_state_dict(gpt_sd)
        self.bigvgan.load_state_dict(bigvgan_sd)
        self.text_embedding.copy_(gpt_sd["text_embedding.weight"])
        self.text_pos_embedding.copy_(gpt_sd["text_pos_embedding.emb.weight"])
        self.speed_emb.copy_(gpt_sd["speed_emb.weight"])
        return self

    def broadcast_tensors(self):
        """Device tensors that together hold every weight (rank 0 -> all, one call each)."""
        gp, gn = self.gpt.arena()
        bp, bn = self.bigvgan.arena()
        return [arena_tensor(gp, gn, self.device), arena_tensor(bp, bn, self.device), self.text_embedding,
                self.text_pos_embedding, self.speed_emb]

    def adopt(self):
        self.gpt.adopt_arena()
        self.bigvgan.adopt_arena()
        return self

    # ------------------------------------------------------------------ G0
    def prepare_gpt_inputs(self, conds_latent, text_ids):
        """-> (embeds [P-1, D], n_left_pad, P); see the module-level `prepare_gpt_inputs`."""
        return prepare_gpt_inputs(self.gpt_cfg, self.text_embedding, self.text_pos_embedding, conds_latent, text_ids)

    def conds_latent(self, cond32, emo_vec):
        """inference_speech (model_v2.py:693-696)."""
        c = cond32.to(self.device, torch.float32) + emo_vec.to(self.device, torch.float32).reshape(1, -1)
        return torch.cat((c, self.speed_emb[1:2], self.speed_emb[0:1]), 0)

    def latent_prefix(self, conds_latent, text_ids):
        """[conds ; text_emb] rows of UnifiedVoice.forward (model_v2.py:579-589)."""
        c = self.gpt_cfg
        t = torch.as_tensor(text_ids, dtype=torch.long, device=self.device).reshape(-1)
        t = torch.cat((t.new_tensor([c["start_text_token"]]), t, t.new_tensor([c["stop_text_token"]])))
        temb = self.text_embedding[t] + self.text_pos_embedding[: t.numel()]
        return torch.cat((conds_latent.to(self.device, torch.float32), temb), 0)

    # ------------------------------------------------------------------ G1-G8
    def generate(self, prompts, max_new, repetition_penalty=10.0, fixed_length=False, sync_every=64, **sampler):
        """Decode up to `max_batch` independent sequences together: greedy by default, multinomial sampling with
        `do_sample=True, temperature=, top_k=, top_p=, seed=` (each slot draws from its own counter-based stream).

        prompts: list of (embeds [P-1,D], n_left_pad).  Returns a list of int32 id arrays,
        trimmed at the first stop token (inclusive), as `generate()` would return them.
        """
        B = len(prompts)
        assert 1 <= B <= self.gpt.max_batch
        for b, (emb, pad) in enumerate(prompts):
            self.gpt.prefill(b, emb, pad)
        done = 0
        out = [None] * B
        while done < max_new:
            n = min(sync_every, max_new - done)
            self.gpt.decode(B, n, repetition_penalty=repetition_penalty, suppress_stop=fixed_length, **sampler)
            done += n
            fins = []
            for b in range(B):
                ids, fin = self.gpt.read(b)
                out[b] = ids[:max_new]
                fins.append(fin)
            if all(fins):
                break
        return out

    def generate_many(self, segments, fixed_length=False, repetition_penalty=10.0, sync_every=64):
        """Row N3: decode any number of segments (of any number of requests) with continuous batching over the engine's
        slots.  segments: list of (embeds [P-1,D], n_left_pad, max_new).  Returns the id arrays in submission order."""
        from .scheduler import DecodeScheduler, Segment

        out = [None] * len(segments)
        sched = DecodeScheduler(self.gpt, self.gpt.max_batch, self.gpt_cfg["stop_mel_token"], sync_every=sync_every)
        segs = [Segment(0, i, e, p, n) for i, (e, p, n) in enumerate(segments)]
        self.last_sched_stats = sched.run(segs, lambda seg, ids: out.__setitem__(seg.index, ids), fixed_length=fixed_length,
                                          repetition_penalty=repetition_penalty)
        return out

    def generate_beams_many(self, segments, num_beams=3, fixed_length=False, sync_every=64, repetition_penalty=10.0, temperature=0.8,
                            top_k=30, top_p=0.8, seed=0, length_penalty=0.0, typical_mass=0.0):
        """The served default (`num_beams=3` beam-sample, infer_v2.py:598-606) for any number of segments: each segment is a
        beam group of `num_beams` slots, the engine's groups step together (a wide engine holds floor(max_batch / num_beams);
        up to 4 slots: one group, the segments in turn as the reference runs them).  segments: list of (embeds [P-1,D],
        n_left_pad, max_new).  Returns the best hypothesis per segment, in submission order."""
        from .scheduler import BeamGroupScheduler, Segment

        out = [None] * len(segments)
        sched = BeamGroupScheduler(self.gpt, num_beams, sync_every=sync_every)
        segs = [Segment(0, i, e, p, n) for i, (e, p, n) in enumerate(segments)]
        self.last_sched_stats = sched.run(segs, lambda seg, ids, score: out.__setitem__(seg.index, ids), fixed_length=fixed_length,
                                          repetition_penalty=repetition_penalty, temperature=temperature, top_k=top_k, top_p=top_p, seed=seed,
                                          length_penalty=length_penalty, typical_mass=typical_mass)
        return out

    # ------------------------------------------------------------------ G9
    def latent(self, conds_latent, text_ids, codes):
        prefix = self.latent_prefix(conds_latent, text_ids)
        codes = torch.as_tensor(np.asarray(codes), dtype=torch.int32, device=self.device)
        return self.gpt.latent(prefix, codes)

    # ------------------------------------------------------------------ N2 (PyTorch glue, optional)
    def attach_conditioning(self, W, cfg=None):
        """Conformer/perceiver conditioning encoders (voice-tts_amd/conditioning.py); W keyed like UnifiedVoice.state_dict()."""
        from .conditioning import COND_CFG, Conditioning

        self.cond_model = Conditioning(W, COND_CFG if cfg is None else cfg, device=self.device)
        return self

    @torch.no_grad()
    def conds_from_prompt(self, spk_cond_emb, emo_cond_emb=None, emo_alpha=1.0):
        """infer_v2.py:629-635 + model_v2.py:684-696: w2v-bert features [1,T,1024] of the speaker (and emotion) prompt ->
        conds_latent [34, D].  Once per request (the reference recomputes it per segment)."""
        cond32, emovec = self.cond_model.encode_prompt(spk_cond_emb, emo_cond_emb, emo_alpha)
        return self.conds_latent(cond32, emovec)

    # ------------------------------------------------------------------ N1 (PyTorch glue, optional)
    def attach_s2mel(self, W, cfg=None):
        """Semantic-to-mel stage (length regulator + CFM/DiT) as PyTorch-ROCm glue on this device (voice-tts_amd/s2mel.py)."""
        from .s2mel import S2MEL_CFG, S2Mel

        self.s2mel_model = S2Mel(W, S2MEL_CFG if cfg is None else cfg, device=self.device)
        return self

    def s2mel(self, latent, codes, prompt_condition, ref_mel, style, n_timesteps=25, inference_cfg_rate=0.7, noise=None):
        """infer_v2.py:713-731: latent [n,D] + codes [n] -> mel [1,80,floor(1.72 n)] (fp32, as the reference runs this stage)."""
        codes = torch.as_tensor(np.asarray(codes), dtype=torch.long, device=self.device).reshape(1, -1)
        lens = torch.tensor([codes.shape[1]], device=self.device)
        return self.s2mel_model(latent.reshape(1, codes.shape[1], -1), codes, lens, prompt_condition, ref_mel, style,
                                n_timesteps=n_timesteps, inference_cfg_rate=inference_cfg_rate, noise=noise)

    # ------------------------------------------------------------------ V0-V5
    def vocode(self, mel):
        """bigvgan(mel.float()) then the PCM clamp of infer_v2.py:735-744: returns fp32 [1, T] scaled to +-32767."""
        wav = self.bigvgan(mel.to(self.device, torch.float32)).squeeze(1)
        return torch.clamp(32767 * wav, -32767.0, 32767.0)


def audio_seconds(n_codes_per_segment, interval_silence_ms=200):
    """Audio length the reference produces for these code counts (infer_v2.py:719,752-754)."""
    total = 0.0
    for n in n_codes_per_segment:
        frames = int(n * 1.72)
        total += frames * 256 / SAMPLE_RATE
    total += (len(n_codes_per_segment) - 1) * interval_silence_ms / 1000.0
    return total
