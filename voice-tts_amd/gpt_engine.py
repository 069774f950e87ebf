"""Seam 1: the decode engine that stands where DeepSpeed's injected model stands.

Mirrors the surface `UnifiedVoice.inference_speech` uses on `self.inference_model`
(indextts/gpt/model_v2.py:698,724-729): `store_mel_emb(embeds)` then
`generate(inputs, bos_token_id, pad_token_id, eos_token_id, attention_mask, max_length,
logits_processor, num_return_sequences, **hf_generate_kwargs)` returning a LongTensor
`[num_return_sequences, P + n]`; plus the latent forward of `UnifiedVoice.forward`
(model_v2.py:554-596).  All arithmetic runs in libixtts_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .weights import GPT_CFG

GPT_TENSOR_PREFIXES = ("gpt.h.", "gpt.ln_f.", "final_norm.", "mel_head.", "mel_embedding.", "mel_pos_embedding.")


class GptEngine:
    def __init__(self, cfg=None, dtype="f32", max_seq=2048, max_batch=1, device=None):
        cfg = dict(GPT_CFG if cfg is None else cfg)
        self.cfg = cfg
        self.device = torch.device(device if device is not None else "cuda:0")
        self.dtype = dtype
        c = _lib.GptCfg()
        c.model_dim = cfg["model_dim"]
        c.layers = cfg["layers"]
        c.heads = cfg["heads"]
        c.n_mel_codes = cfg["number_mel_codes"]
        c.n_mel_pos = cfg["max_mel_tokens"] + 3
        c.n_text_tokens = cfg["number_text_tokens"] + 1
        c.n_text_pos = cfg["max_text_tokens"] + 2
        c.start_mel_token = cfg["start_mel_token"]
        c.stop_mel_token = cfg["stop_mel_token"]
        c.max_seq = max_seq
        c.max_batch = max_batch
        c.weight_dtype = {"f32": 0, "bf16": 1}[dtype]
        self._c = c
        self.max_seq = max_seq
        self.max_batch = max_batch
        self.D = cfg["model_dim"]
        self.V = cfg["number_mel_codes"]
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ixtts_gpt_create(C.byref(self._h), C.byref(c)), "ixtts_gpt_create")
        self._loaded = False
        self.cached_mel_emb = None

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd):
        """Feed the engine from a `UnifiedVoice.state_dict()`-style mapping (fp32 host tensors)."""
        L = _lib.lib()
        with torch.cuda.device(self.device):
            for name, t in sd.items():
                if not name.startswith(GPT_TENSOR_PREFIXES):
                    continue
                t = t.detach().to("cpu", torch.float32).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                _lib.check(L.ixtts_gpt_set_tensor(self._h, name.encode(), t.data_ptr(), shape, t.dim()), f"ixtts_gpt_set_tensor({name})")
            _lib.check(L.ixtts_gpt_finalize(self._h), "ixtts_gpt_finalize")
        self._loaded = True
        return self

    def arena(self):
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().ixtts_gpt_arena(self._h, C.byref(p), C.byref(n)), "ixtts_gpt_arena")
        return p.value, n.value

    def adopt_arena(self):
        _lib.check(_lib.lib().ixtts_gpt_adopt_arena(self._h), "ixtts_gpt_adopt_arena")
        self._loaded = True

    def share_arena(self, owner):
        """Read `owner`'s device weights instead of holding a copy (same model, same dtype): a second engine shape -- e.g. the
        wide beam-group engine beside the register engine -- costs its KV cache only."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ixtts_gpt_share_arena(self._h, owner._h), "ixtts_gpt_share_arena")
        self._owner = owner  # keep it alive
        self._loaded = True
        return self

    # ------------------------------------------------------------------ low level
    def _stream(self):
        return _lib.current_stream_ptr()

    def prefill(self, slot, embeds, n_left_pad=0):
        """embeds [P-1, D] fp32 (cond + text rows, left-padded with zero rows)."""
        e = embeds.to(self.device, torch.float32).contiguous()
        assert e.dim() == 2 and e.shape[1] == self.D
        with torch.cuda.device(self.device):
            rc = _lib.lib().ixtts_gpt_prefill(self._h, slot, e.data_ptr(), e.shape[0], int(n_left_pad), self._stream())
        _lib.check(rc, "ixtts_gpt_prefill")
        self._keep = e  # keep alive until the stream has consumed it

    def decode(self, n_active, n_steps, repetition_penalty=10.0, temperature=1.0, top_k=0, top_p=1.0,
               do_sample=False, suppress_stop=False, seed=0, typical_mass=0.0):
        sc = _lib.SamplerCfg(repetition_penalty, temperature, top_k, top_p, int(do_sample), int(suppress_stop), seed, float(typical_mass), 0.0)
        with torch.cuda.device(self.device):
            rc = _lib.lib().ixtts_gpt_decode(self._h, n_active, n_steps, C.byref(sc), self._stream())
        _lib.check(rc, "ixtts_gpt_decode")

    def read(self, slot):
        ids = np.zeros(self.max_seq, dtype=np.int32)
        n, fin = C.c_int(), C.c_int()
        with torch.cuda.device(self.device):
            rc = _lib.lib().ixtts_gpt_read(self._h, slot, ids.ctypes.data, ids.size, C.byref(n), C.byref(fin), self._stream())
        _lib.check(rc, "ixtts_gpt_read")
        return ids[: n.value].copy(), bool(fin.value)

    def read_logits(self, slot):
        out = np.zeros(self.V, dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ixtts_gpt_read_logits(self._h, slot, out.ctypes.data, self._stream()), "ixtts_gpt_read_logits")
        return out

    def read_probs(self, slot):
        out = np.zeros(self.V, dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ixtts_gpt_read_probs(self._h, slot, out.ctypes.data, self._stream()), "ixtts_gpt_read_probs")
        return out

    def force_next(self, slot, token):
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ixtts_gpt_force_next(self._h, slot, int(token), self._stream()), "ixtts_gpt_force_next")

    # ------------------------------------------------------------------ beam-sample
    # Group g = the beams of one prompt (prefilled into slot g * num_beams); groups step together on a wide engine.
    def beam_begin(self, num_beams, group=0, rng_stream=0):
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ixtts_gpt_beam_begin_group(self._h, int(group), int(num_beams), int(rng_stream), self._stream()), "ixtts_gpt_beam_begin_group")
        self._nb = int(num_beams)

    def beam_park(self, group):
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ixtts_gpt_beam_park_group(self._h, int(group), self._stream()), "ixtts_gpt_beam_park_group")

    def beam_decode(self, n_steps, repetition_penalty=10.0, temperature=0.8, top_k=30, top_p=0.8, suppress_stop=False, seed=0, typical_mass=0.0,
                    length_penalty=0.0, groups=1, do_sample=True):
        """`do_sample=False`: beam search proper -- the joint top 2 * num_beams instead of the multinomial draw, no warpers."""
        sc = _lib.SamplerCfg(repetition_penalty, temperature, top_k, top_p, int(bool(do_sample)), int(suppress_stop), seed, float(typical_mass), float(length_penalty))
        self._lp = float(length_penalty)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ixtts_gpt_beam_decode_groups(self._h, int(groups), n_steps, C.byref(sc), self._stream()), "ixtts_gpt_beam_decode_groups")

    def beam_force(self, picks, group=0):
        a = np.ascontiguousarray(np.asarray(picks, dtype=np.int32))
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ixtts_gpt_beam_force_group(self._h, int(group), a.ctypes.data, a.size, self._stream()), "ixtts_gpt_beam_force_group")

    def beam_read(self, max_new, group=0):
        ids = np.zeros(self.max_seq + 1, dtype=np.int32)
        n, done, score = C.c_int(), C.c_int(), C.c_float()
        bs = np.zeros(self._nb, np.float32)
        lt = np.zeros(self._nb, np.int32)
        src = np.zeros(self._nb, np.int32)
        with torch.cuda.device(self.device):
            rc = _lib.lib().ixtts_gpt_beam_read_group(self._h, int(group), int(max_new), ids.ctypes.data, ids.size, C.byref(n), C.byref(done), C.byref(score),
                                                      bs.ctypes.data, lt.ctypes.data, src.ctypes.data, self._stream())
        _lib.check(rc, "ixtts_gpt_beam_read_group")
        return ids[: n.value].copy(), bool(done.value), float(score.value), bs, lt, src

    def latent(self, prefix, codes):
        """prefix [34+L+2, D] fp32 (conds ; text_emb); codes int [n] -> latent [n, D] (model_v2.py:554-596)."""
        p = prefix.to(self.device, torch.float32).contiguous()
        c = codes.to(self.device, torch.int32).contiguous()
        out = torch.empty(c.numel(), self.D, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            rc = _lib.lib().ixtts_gpt_latent(self._h, p.data_ptr(), p.shape[0], c.data_ptr(), c.numel(), out.data_ptr(), self._stream())
        _lib.check(rc, "ixtts_gpt_latent")
        self._keep2 = (p, c)
        return out

    def step_bytes(self, B, S):
        return _lib.lib().ixtts_gpt_step_bytes(self._h, B, S)

    def bench_gemv(self, which, layer, batch=1):
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ixtts_gpt_bench_gemv(self._h, which, layer, batch, self._stream()), "ixtts_gpt_bench_gemv")

    # ------------------------------------------------------------------ reference-shaped surface
    def store_mel_emb(self, mel_emb):
        """GPT2InferenceModel.store_mel_emb (model_v2.py:87-88)."""
        self.cached_mel_emb = mel_emb

    def generate(self, inputs, bos_token_id=None, pad_token_id=None, eos_token_id=None, attention_mask=None,
                 max_length=None, logits_processor=None, num_return_sequences=1, do_sample=True, top_p=1.0, top_k=0,
                 temperature=1.0, num_beams=1, repetition_penalty=1.0, length_penalty=0.0, sync_every=64,
                 suppress_stop=False, **unused):
        """The slice of HF `generate()` that `inference_speech` exercises (model_v2.py:724-729).

        `inputs` = fake ids [1, P] (only its length matters, model_v2.py:652-661); the
        prompt rows come from `store_mel_emb`.  Greedy == `top_k=1` or `do_sample=False`
        (SURVEY.md F3).  Host syncs once per `sync_every` steps (finished flag), not per token.
        """
        if self.cached_mel_emb is None:
            raise RuntimeError("generate(): call store_mel_emb first (model_v2.py:137)")
        if num_beams != 1 and not (2 <= num_beams <= min(self.max_batch, 4)):
            raise NotImplementedError("beam mode needs 2 <= num_beams <= min(max_batch, 4) (beam-sample, the served configuration, or with do_sample=False beam search)")
        # the one custom processor inference_speech ever builds is TypicalLogitsWarper(mass=typical_mass) (model_v2.py:717-722):
        # it runs on the device; anything else has no kernel
        typical_mass = float(unused.pop("typical_mass", 0.0)) if unused.pop("typical_sampling", False) else 0.0
        for proc in (logits_processor or []):
            if type(proc).__name__ == "TypicalLogitsWarper" and hasattr(proc, "mass"):
                typical_mass = float(proc.mass)
            else:
                raise NotImplementedError(f"logits processor {type(proc).__name__} has no device implementation (only typical sampling does)")
        if inputs.shape[0] != 1:
            raise NotImplementedError("one prompt per generate() call (autoregressive_batch_size = 1, infer_v2.py:602)")
        nret = int(num_return_sequences)
        if nret != 1 and (num_beams != 1 or not (1 <= nret <= self.max_batch)):
            raise NotImplementedError("num_return_sequences > 1 is built for sampling without beams, up to max_batch sequences "
                                      "(HF expands the prompt and draws independent continuations, generation_utils.py:2128-2135)")
        greedy = (not do_sample) or top_k == 1
        if num_beams != 1 and do_sample and not (1 <= top_k <= 128):
            raise NotImplementedError("beam-sample keeps at most 128 candidates per beam on the device: 1 <= top_k <= 128 (sampling without beams takes any top_k, 0 = off)")
        top_k = max(0, int(top_k))
        P = inputs.shape[1]
        emb = self.cached_mel_emb
        emb = emb[0] if emb.dim() == 3 else emb
        assert emb.shape[0] == P - 1, (emb.shape, P)
        n_pad = 0
        if attention_mask is not None:
            m = attention_mask.reshape(-1)
            n_pad = int((m == 0).sum().item())
            assert n_pad == 0 or bool((m[:n_pad] == 0).all()), "only left padding is produced by prepare_gpt_inputs (model_v2.py:639-642)"
        max_new = (max_length - P) if max_length is not None else (self.max_seq - P - 2)
        # the mel position table bounds what can be embedded (inference_speech's own default cap: max_mel_tokens - 1, model_v2.py:699-703)
        max_new = max(0, min(max_new, self.max_seq - P - 2, self.cfg["max_mel_tokens"] - 1))
        self.prefill(0, emb, n_pad)
        if num_beams != 1:
            # served default: 3-beam beam-sample (infer_v2.py:598-605,641-658); do_sample=False: beam search (generation_utils.py:3520-3524)
            self.beam_begin(num_beams)
            done_steps, fin = 0, False
            ids = np.zeros(0, np.int32)
            while done_steps < max_new and not fin:
                n = min(sync_every, max_new - done_steps)
                self.beam_decode(n, repetition_penalty=repetition_penalty, temperature=temperature, top_k=top_k, top_p=top_p,
                                 suppress_stop=suppress_stop, seed=int(unused.get("seed", 0)), typical_mass=typical_mass, length_penalty=length_penalty,
                                 do_sample=do_sample)
                done_steps += n
                ids, fin = self.beam_read(max_new)[:2]
            out = torch.cat([inputs.reshape(1, -1).to(torch.long).cpu(), torch.from_numpy(ids.astype(np.int64)).reshape(1, -1)], dim=1)
            return out.to(inputs.device)
        if nret > 1:
            # `input_ids.repeat_interleave(num_return_sequences)`: the same prompt in nret slots, each drawing from its own stream
            # (the slot index is part of the counter-based RNG); rows are right-padded with pad_token_id as HF pads finished rows
            for b in range(1, nret):
                self.prefill(b, emb, n_pad)
            done, fins = 0, [False] * nret
            rows = [np.zeros(0, np.int32)] * nret
            while done < max_new and not all(fins):
                n = min(sync_every, max_new - done)
                self.decode(nret, n, repetition_penalty=repetition_penalty, temperature=temperature, top_k=top_k, top_p=top_p,
                            do_sample=not greedy, suppress_stop=suppress_stop, seed=int(unused.get("seed", 0)), typical_mass=typical_mass)
                done += n
                for b in range(nret):
                    rows[b], fins[b] = self.read(b)
            rows = [r[:max_new] for r in rows]
            width = max(len(r) for r in rows)
            pad = int(pad_token_id if pad_token_id is not None else self.cfg["stop_mel_token"])
            body = np.full((nret, width), pad, np.int64)
            for b, r in enumerate(rows):
                body[b, : len(r)] = r
            out = torch.cat([inputs.reshape(1, -1).to(torch.long).cpu().repeat(nret, 1), torch.from_numpy(body)], dim=1)
            return out.to(inputs.device)
        done = 0
        ids, fin = np.zeros(0, np.int32), False
        while done < max_new and not fin:
            n = min(sync_every, max_new - done)
            self.decode(1, n, repetition_penalty=repetition_penalty, temperature=temperature, top_k=top_k, top_p=top_p,
                        do_sample=not greedy, suppress_stop=suppress_stop, seed=int(unused.get("seed", 0)), typical_mass=typical_mass)
            done += n
            ids, fin = self.read(0)
        ids = ids[:max_new]
        out = torch.cat([inputs.reshape(1, -1).to(torch.long).cpu(), torch.from_numpy(ids.astype(np.int64)).reshape(1, -1)], dim=1)
        return out.to(inputs.device)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                _lib.lib().ixtts_gpt_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass
