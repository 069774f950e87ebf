#!/usr/bin/env python
"""bench.py -- throughput of the MI355X hot path on BASELINE.json's metric (RTF / audio-seconds per second).

Default workload (BASELINE configs[1], SURVEY.md 8(d) "Config 2"): ONE /tts request per GPU = 200-token zh text ->
2 segments x 100 tokens (prompt P = 137 rows), 5 s speaker prompt -> the conditioning encoders (conformer + perceiver, torch
glue) -> greedy fixed-length decode of n = 1100 codes per segment, both segments decoded together (stop token suppressed:
random weights never emit EOS) -> the latent GPT forward per segment -> s2mel (length regulator + 25-step CFM/DiT, torch glue
with HIP attention / row kernels) -> BigVGAN over floor(1.72 n) = 1892 mel frames per segment -> 44.13 s of audio.  (s2mel and
BigVGAN are fp32 end to end; their attention / conv products run as six exact bf16 MFMA partial products per fp32 product with fp32
accumulation, DESIGN.md 4.4 -- IXTTS_ATTN_FULL=f32 / IXTTS_BV_CONV=f32 select the fp32-MFMA kernels.)  ALL of
these stages are inside the timed region; what stays synthetic is what the reference caches per speaker prompt (w2v-bert
features, CAM++ style, prompt condition, reference mel): seeded HBM-resident tensors of the production shapes.

One "step" = one such request per rank.  Weak scaling: every rank serves its own request (process-per-GPU sharding,
gunicorn_config.py:43-60); RCCL only broadcasts the weights from rank 0 at load (the packed GPT / BigVGAN arenas and one packed
buffer of the glue weights -- voice-tts_amd/sharding.py).  `--gpus N` without a torchrun environment starts its own N rank
processes (a child `python -m torch.distributed.run`, before this process touches a GPU).

`--workload mixed64` is BASELINE configs[3] (SURVEY config 4): 64 requests of 50-400 characters with distinct prompts,
request i -> rank i mod N, each rank's segments decoded through the continuous-batching scheduler; the whole job is one step
(strong scaling).  `--decode beam`, `--dtype f32`, `--bigvgan-only`, `--concurrency R` select the other BASELINE configs.

The default N=1 line also carries, under "extra", figures the driver would otherwise never see: the config-5 BigVGAN
microbench, the served-default 3-beam request, the fp32 (parity-mode) request and the decode step per batch size.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

PEAK_HBM, PEAK_F32, PEAK_BF16 = 8000.0, 157.3, 2500.0  # GB/s, TFLOP/s fp32 MFMA, TFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md)


def conv_roofline(tflops):
    """Roofline block of the BigVGAN convs.  Default build: fp32-accurate products as six bf16 MFMAs of three-way split operands
    (csrc/conv1d_x3.hip) -- the bound is the dense bf16 MFMA peak / 6 per fp32-equivalent flop; IXTTS_BV_CONV=f32: the fp32 MFMA peak.
    `tflops` counts 2 * Cout * Cin * k * T per conv either way (fp32-equivalent), so the fp32-MFMA figure stays comparable."""
    x3 = os.environ.get("IXTTS_BV_CONV", "x3") != "f32"
    peak = PEAK_BF16 / 6 if x3 else PEAK_F32
    return {"bound": "mfma", "achieved": round(tflops, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tflops / peak, 3),
            "arithmetic": "fp32-equivalent flops; six bf16 MFMA partial products per fp32 product, fp32 accumulate" if x3 else "fp32 MFMA",
            "fp32_mfma_peak": PEAK_F32, "frac_of_fp32_mfma_peak": round(tflops / PEAK_F32, 3)}


def host_cores():
    """CPU share actually available to this process (cgroup quota / affinity), not the host's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("IXTTS_CPU_THREADS", "16"))))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="request", choices=["request", "mixed64"], help="request: BASELINE configs[1] (the judged line); "
                    "mixed64: configs[3] -- 64 mixed-length requests, request i -> rank i mod N, continuous batching on each rank")
    ap.add_argument("--tokens", type=int, default=100, help="text tokens per segment")
    ap.add_argument("--codes", type=int, default=1100, help="mel codes per segment (11 per char)")
    ap.add_argument("--segments", type=int, default=2)
    ap.add_argument("--slots", type=int, default=0, help="decode slots for --concurrency / mixed64 (0 = the engine's maximum)")
    ap.add_argument("--no-s2mel", action="store_true", help="leave the PyTorch-glue s2mel stage out of the timed region (feed synthetic mels)")
    ap.add_argument("--concurrency", type=int, default=1, help="requests in flight per GPU: their segments share the decode slots through the "
                    "continuous-batching scheduler (row N3); 1 = BASELINE configs[1], the judged line")
    ap.add_argument("--no-cond", action="store_true", help="leave the conditioning encoders out of the timed region (feed a synthetic conds_latent)")
    ap.add_argument("--decode", default="greedy", choices=["greedy", "beam", "beam-turn", "sample"], help="greedy: BASELINE configs[1] (the judged line); beam: the "
                    "served default -- 3-beam beam-sample, top_k 30, top_p 0.8, temperature 0.8 -- every segment a beam group, the groups stepping together "
                    "on the wide engine (what infer() runs); beam-turn: the same, one segment at a time on the register engine (the reference's order); "
                    "sample: BASELINE configs[2] -- top-p sampling without beams (top_p 0.8, top_k 30, temperature 0.8), both segments together")
    ap.add_argument("--emo-alpha", type=float, default=None, help="BASELINE configs[2]: a separate emotion prompt (5 s, 249 w2v-bert frames) merged with this "
                    "alpha (0.7) in the conditioning encoders (merge_emovec, model_v2.py:742-747)")
    ap.add_argument("--bigvgan-only", action="store_true", help="BASELINE configs[4]: 1000-frame random mel -> waveform microbench (20 warm-up + 100 timed)")
    ap.add_argument("--no-prompt-side", action="store_true", help="mixed64: feed synthetic per-prompt features instead of running the prompt-side models "
                    "(w2v-bert, codec, CAM++, mel) on a distinct 5 s recording per request")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-load-warmup", action="store_true", help="skip the short synthetic request the loader runs before the first real one (server lifespan does the same)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra figures of the default line (config 5, beam, fp32, step per batch size)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N rank processes ourselves.  This process has not touched a GPU
    (importing torch does not), it only waits for the child launcher and passes its output and exit code on."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting ranks: " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ================================================================================================ stage microbenches
def bigvgan_microbench(WR, dev, model=None):
    """BASELINE configs[4] / SURVEY 8(d) config 5: mel ~ N(-4, 2) clipped to [-11.5, 2], fp32 [1, 80, 1000] -> [1, 1, 256000]."""
    from voice_tts_amd.bigvgan import BigVGAN

    m = model if model is not None else BigVGAN(WR.BIGVGAN_CFG, max_frames=1024, device=dev).load_state_dict(WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234))
    F = 1000
    mel = (torch.randn(1, 80, F, generator=torch.Generator().manual_seed(6)) * 2 - 4).clamp(-11.5, 2).to(dev)
    for _ in range(20):
        m(mel)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        m(mel)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    fl = m.flops(1, F)
    alg_bytes = 732.8e6  # weights once + stage-boundary activations once + mel + wav, per 1000 frames
    return {"metric": "bigvgan_ms_per_1000_frames", "value": round(ms, 3), "unit": "ms", "n_gpus": 1, "steps": 100, "warmup": 20,
            "higher_is_better": False, "dtype": "f32", "data": "synthetic", "audio_seconds_per_second": round(256 * F / 22050 / (ms * 1e-3), 1),
            "config": {"workload": "BigVGAN-only: 1000-frame random mel -> 256000 samples (11.61 s)"},
            "roofline": dict(conv_roofline(fl / ms / 1e9), flops=fl, hbm_GBps_algorithmic=round(alg_bytes / (ms * 1e-3) / 1e9, 1), hbm_peak_GBps=PEAK_HBM)}


def decode_step_by_batch(hp, P, batches, n_steps=256):
    """Decode step time per batch size B (weights are read once for B sequences): B prompts of P rows, `n_steps` greedy steps,
    HIP events on the launch stream.  Context runs from P to P + n_steps."""
    out = {}
    D = hp.gpt.D
    emb = (torch.randn(P - 1, D, generator=torch.Generator().manual_seed(1)) * 0.5).to(hp.device)
    for B in batches:
        if B > hp.gpt.max_batch:
            continue
        for b in range(B):
            hp.gpt.prefill(b, emb, 0)
        hp.gpt.decode(B, 16, repetition_penalty=10.0, suppress_stop=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hp.gpt.decode(B, n_steps, repetition_penalty=10.0, suppress_stop=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n_steps
        by = hp.gpt.step_bytes(B, P + 16 + n_steps // 2)
        out[str(B)] = {"us": round(us, 1), "alg_bytes": by, "achieved_GBps": round(by / us / 1e3, 1), "frac": round(by / us / 1e3 / PEAK_HBM, 4),
                       "us_per_sequence": round(us / B, 1)}
    return out


def build_prompt_encoder(hp, dev):
    """The once-per-NEW-prompt stages (infer_v2.py:508-545: audio decode, resampling, w2v-bert features, semantic codec, reference
    mel, kaldi fbank + CAM++, prompt condition through the s2mel length regulator -- voice-tts_amd/prompt.py) at PRODUCTION size
    with seeded random weights: `Wav2Vec2BertModel(Wav2Vec2BertConfig())` is the w2v-bert-2.0 shape (24 layers x 1024, 580 M
    parameters; the installed transformers class, as the reference takes it from that library), RepCodec at CODEC_CFG, CAM++."""
    import voice_tts_amd.prompt as PR
    from transformers import Wav2Vec2BertConfig, Wav2Vec2BertModel

    torch.manual_seed(1234)
    w2v = PR.W2vBert(Wav2Vec2BertModel(Wav2Vec2BertConfig()), torch.zeros(1024), torch.ones(1024), device=dev)
    codec = PR.SemanticCodec(PR.make_codec_weights(PR.CODEC_CFG, seed=1234), PR.CODEC_CFG, dev)
    cam = PR.CamPlus(PR.make_camplus_weights(seed=1234), dev)
    return PR.PromptEncoder(w2v, codec, cam, hp.s2mel_model, dev)


def prompt_wav(seconds=5.0, sr=24000, seed=0):
    """A mono 16-bit WAVE byte string (harmonics + noise): what `/tts` hands over as `spk_audio` (server.py:352-370)."""
    import io
    import wave

    rng = np.random.RandomState(seed)
    t = np.arange(int(seconds * sr)) / sr
    x = sum(a * np.sin(2 * np.pi * f * (1 + 0.01 * seed) * t + p) for a, f, p in ((0.3, 140, 0), (0.2, 280, 1), (0.1, 420, 2), (0.05, 1900, 3)))
    x = (x * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t)) + 0.02 * rng.randn(t.size)).astype(np.float32)
    b = io.BytesIO()
    with wave.open(b, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
    return b.getvalue()


# ================================================================================================ the request
class Workload:
    """Synthetic inputs of one rank, resident in HBM before the timed region, and the request loop over the HotPath."""

    def __init__(self, args, hp, dev, rank, use_s2mel, use_cond):
        import voice_tts_amd.weights as WR

        self.args, self.hp, self.dev, self.use_s2mel, self.use_cond = args, hp, dev, use_s2mel, use_cond
        self.D = WR.GPT_CFG["model_dim"]
        self.Tref = 430  # 5 s speaker prompt -> 430 reference mel frames, 249 w2v-bert frames
        self.g = torch.Generator().manual_seed(100 + rank)
        self.stage_ms = {"gpt_gen": 0.0, "gpt_forward": 0.0, "s2mel": 0.0, "bigvgan": 0.0}
        self.prompt_enc = None  # mixed64 / extra.prompt_side: the prompt-side models at production size (build_prompt_encoder)

    def prompt_from_audio(self, wav_bytes):
        """What a request with a NEW speaker prompt pays before its first segment (infer_v2.py:508-545), on the device."""
        d = self.prompt_enc.speaker(wav_bytes)
        # (random-weight w2v-bert features are not unit-variance; the conditioning encoders and the DiT are fed at the scale they are built for)
        d["spk_cond_emb"] = d["spk_cond_emb"] / d["spk_cond_emb"].std().clamp_min(1e-6)
        d["emo_cond_emb"] = d["spk_cond_emb"]
        d["conds"] = None
        return d

    def prompt(self):
        """Stand-ins for what the reference caches per speaker prompt (infer_v2.py:508-545)."""
        g, dev = self.g, self.dev
        return dict(spk_cond_emb=torch.randn(1, 249, 1024, generator=g).to(dev), prompt_condition=torch.randn(1, self.Tref, 512, generator=g).to(dev),
                    ref_mel=(torch.randn(1, 80, self.Tref, generator=g) * 2 - 4).clamp(-11.5, 2).to(dev), style=torch.randn(1, 192, generator=g).to(dev),
                    emo_cond_emb=torch.randn(1, 249, 1024, generator=g).to(dev), conds=(torch.randn(34, self.D, generator=g) * 0.5).to(dev))

    def conds(self, pr):
        if not self.use_cond:
            return pr["conds"]
        ea = self.args.emo_alpha
        cl = self.hp.conds_from_prompt(pr["spk_cond_emb"], pr["emo_cond_emb"] if ea is not None else None, ea if ea is not None else 1.0)  # merge_emovec + get_conditioning (infer_v2.py:629-635), once per request
        # synthetic encoder weights give arbitrary latent statistics; keep the GPT prefix at the scale it is built for
        return cl * (0.5 / cl.std().clamp_min(1e-6))

    def post(self, hp, pr, cl, text, codes, acc):
        """latent pass -> s2mel -> BigVGAN -> int16 on the host, one segment; stage times accumulate in `acc`."""
        tick = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
        ta = tick()
        lat = hp.latent(cl, text, codes)
        tb = tick()
        frames = int(len(codes) * 1.72)
        if self.use_s2mel:  # 25 Euler steps x CFG batch 2 over T = 430 + frames, fp32 (infer_v2.py:713-731)
            mel = hp.s2mel(lat, codes, pr["prompt_condition"], pr["ref_mel"], pr["style"]).clamp(-11.5, 2.0)  # synthetic weights: keep the log-mel range
        else:
            mel = (torch.randn(1, 80, frames, generator=torch.Generator().manual_seed(frames)) * 2 - 4).clamp(-11.5, 2).to(self.dev)
        tc = tick()
        wav = hp.vocode(mel).to(torch.int16).cpu()  # wav.cpu() per segment, int16 truncation (infer_v2.py:744,781)
        td = tick()
        assert lat.shape == (len(codes), self.D) and wav.shape[-1] == frames * 256
        acc["gpt_forward"] += (tb - ta) * 1e3
        acc["s2mel"] += (tc - tb) * 1e3
        acc["bigvgan"] += (td - tc) * 1e3


def make_hotpath(args, dev, dtype, max_batch, max_seq, max_frames, share=None):
    from voice_tts_amd.pipeline import HotPath

    hp = HotPath(dtype=dtype, device=dev, max_batch=max_batch, max_seq=max_seq, max_frames=max_frames)
    if share is not None:  # a second decode engine (beam / fp32 figures) over the same vocoder and glue
        hp.bigvgan = share.bigvgan
        hp.text_embedding, hp.text_pos_embedding, hp.speed_emb = share.text_embedding, share.text_pos_embedding, share.speed_emb
        for n in ("s2mel_model", "cond_model"):
            if hasattr(share, n):
                setattr(hp, n, getattr(share, n))
    return hp


def run_request(wl, hp, pr, texts, n_codes, mode, R=1, acc=None):
    """One /tts request (or R of them sharing the decode slots).  mode: greedy | beam."""
    acc = wl.stage_ms if acc is None else acc
    tick = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
    t0 = tick()
    cl = wl.conds(pr)
    if os.environ.get("IXTTS_BENCH_DETAIL"):
        tc_ = tick()
    prompts = [hp.prepare_gpt_inputs(cl, t)[:2] for t in texts]
    if os.environ.get("IXTTS_BENCH_DETAIL"):
        tp_ = tick()
        print(f"[detail] conditioning {1e3 * (tc_ - t0):.2f} ms, prepare_gpt_inputs {1e3 * (tp_ - tc_):.2f} ms", file=sys.stderr)
    if R > 1:
        many = hp.generate_many([(e, p, n_codes) for _ in range(R) for (e, p) in prompts], fixed_length=True, repetition_penalty=10.0)
    elif mode in ("beam", "beam-turn"):
        # served default (infer_v2.py:598-606): each segment a 3-beam group; the engine's groups step together ("beam": wide engine,
        # 3 x segments slots) or the one group of a register engine takes the segments in turn ("beam-turn", the reference's order)
        many = [c[:n_codes] for c in hp.generate_beams_many([(e, p, n_codes) for (e, p) in prompts], num_beams=3, fixed_length=True, repetition_penalty=10.0,
                                                            temperature=0.8, top_k=30, top_p=0.8, seed=7)]
    elif len(prompts) <= hp.gpt.max_batch:
        samp = dict(do_sample=True, temperature=0.8, top_k=30, top_p=0.8, seed=7) if mode == "sample" else {}
        many = hp.generate(prompts, n_codes, repetition_penalty=10.0, fixed_length=True, **samp)
    else:
        many = hp.generate_many([(e, p, n_codes) for (e, p) in prompts], fixed_length=True, repetition_penalty=10.0)
    t1 = tick()
    acc["gpt_gen"] += (t1 - t0) * 1e3
    assert all(len(c) == n_codes for c in many)
    for r in range(R):
        for s, t in enumerate(texts):
            wl.post(hp, pr, cl, t, many[r * len(texts) + s], acc)
    return many


def run_mixed(wl, hp, reqs, prompts, texts, acc):
    """configs[3] on one rank: every request's prompt-side stages (all prompts distinct: the speaker cache never hits), its
    conditioning, all segments through the scheduler, then the post stages."""
    tick = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
    if wl.prompt_enc is not None:
        tp = tick()
        prompts = [wl.prompt_from_audio(w) for w in prompts]  # `prompts` arrive as WAV byte strings
        acc["prompt"] = acc.get("prompt", 0.0) + (tick() - tp) * 1e3
    t0 = tick()
    cls = [wl.conds(pr) for pr in prompts]
    segs, owner = [], []
    for i, toks in enumerate(reqs):
        for s, n in enumerate(toks):
            e, p, _ = hp.prepare_gpt_inputs(cls[i], texts[i][s])
            segs.append((e, p, 11 * n))
            owner.append((i, s))
    codes = hp.generate_many(segs, fixed_length=True, repetition_penalty=10.0)
    acc["gpt_gen"] += (tick() - t0) * 1e3
    for (i, s), c in zip(owner, codes):
        wl.post(hp, prompts[i], cls[i], texts[i][s], c, acc)


# ================================================================================================ main
def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    # torch sizes its CPU pool by the HOST's cores (128 threads on a GPU box whose share is 16): the host-side pieces of a request
    # (WAV decode, resampling, feature extraction, small tensor ops) then run oversubscribed -- 50-100 ms hiccups on 2 ms operations
    torch.set_num_threads(host_cores())  # (per rank: at most 16, IXTTS_CPU_THREADS overrides)

    import voice_tts_amd.weights as WR
    from voice_tts_amd import _lib, sharding
    from voice_tts_amd.pipeline import audio_seconds

    if args.bigvgan_only:
        out = bigvgan_microbench(WR, dev)
        if rank == 0:
            print(json.dumps(out), flush=True)
        return
    D = WR.GPT_CFG["model_dim"]
    n_seg, n_tok, n_codes = args.segments, args.tokens, args.codes
    R = max(1, args.concurrency)
    mixed = args.workload == "mixed64"
    engine_max = _lib.max_batch()
    slots = min(args.slots or engine_max, engine_max)
    if mixed:
        all_reqs = sharding.mixed_requests()
        mine = sharding.my_requests(len(all_reqs), rank, world)  # request i -> rank i mod N (gunicorn_config.py:53-54)
        reqs = [all_reqs[i] for i in mine]
        max_tok = max(max(r) for r in all_reqs)
        P, max_codes, max_batch = 34 + max_tok + 2 + 1, 11 * max_tok, slots
    else:
        P, max_codes = 34 + n_tok + 2 + 1, n_codes
        # R requests in flight: one decode slot per segment (wide engines step up to 16 together on the matrix cores)
        max_batch = min(slots, R * n_seg) if R > 1 else (min(3 * n_seg, engine_max // 3 * 3) if args.decode == "beam" else 3 if args.decode == "beam-turn" else min(max(2, n_seg), engine_max))
    frames = int(max_codes * 1.72)
    hp = make_hotpath(args, dev, args.dtype, max_batch, P + max_codes + 64, frames)

    # ---- load: rank 0 builds the (synthetic, seeded) weights; RCCL broadcasts the packed arenas and ONE packed glue buffer
    t_load = time.time()
    Wg = Wb = None
    use_s2mel, use_cond = not args.no_s2mel, not args.no_cond
    if rank == 0:
        Wg = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
        Wb = WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234)
        hp.load(Wg, Wb)
    glue_shapes, Wglue = [], {}
    if use_s2mel:
        import voice_tts_amd.s2mel as S2

        glue_shapes += S2.s2mel_shapes(S2.S2MEL_CFG)
        if rank == 0:
            Wglue.update(S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234))
    if use_cond:
        import voice_tts_amd.conditioning as CD

        glue_shapes += CD.cond_shapes(CD.COND_CFG)
        if rank == 0:
            Wglue.update(CD.make_cond_weights(CD.COND_CFG, seed=1234))
    if world > 1:
        sharding.broadcast_weights(hp.broadcast_tensors(), src=0)
        if rank != 0:
            hp.adopt()
        Wglue = sharding.broadcast_tensor_dict(Wglue if rank == 0 else None, glue_shapes, dev, src=0, rank=rank)
    if use_s2mel:
        hp.attach_s2mel({k: v for k, v in Wglue.items() if k.split(".")[0] in ("cfm", "length_regulator", "gpt_layer", "quantizer")})
    if use_cond:
        hp.attach_conditioning(Wglue)
    torch.cuda.synchronize()
    t_load = time.time() - t_load
    log(f"weights loaded in {t_load:.1f}s")

    wl = Workload(args, hp, dev, rank, use_s2mel, use_cond)
    # What the server does before /health answers 200 (voice-tts_amd/server.py lifespan -> IndexTTS2.warm_up): one short synthetic
    # request through every stage, so that a fresh process pays its one-off costs (the BLAS library's code objects, first
    # allocations) at LOAD.  Here: ONE segment of the workload's own shape (the library picks its GEMM kernels per shape), i.e. half a
    # request.  Reported as `load_warmup_s`; never part of the timed region.
    t_wu = time.perf_counter()
    if not args.no_load_warmup:
        acc_w = dict(wl.stage_ms)
        pr_w = wl.prompt()
        run_request(wl, hp, pr_w, [torch.randint(2, 12000, (n_tok if not mixed else 100,), generator=wl.g)], n_codes if not mixed else 1100,
                    "greedy" if args.decode in ("greedy", "sample") else args.decode, 1, acc_w)
        torch.cuda.synchronize()
        log(f"load warm-up request done in {time.perf_counter() - t_wu:.2f}s: " + ", ".join(f"{k} {v:.0f} ms" for k, v in acc_w.items()))
    t_wu = time.perf_counter() - t_wu
    if mixed:
        if use_s2mel and use_cond and not args.no_prompt_side:
            log("building the prompt-side models at production size (w2v-bert-2.0 shape, RepCodec, CAM++)")
            wl.prompt_enc = build_prompt_encoder(hp, dev)
            wl.stage_ms["prompt"] = 0.0
            prompts = [prompt_wav(5.0, 24000, seed=1000 * rank + i) for i in range(len(reqs))]  # distinct 5 s recordings (infer_v2.py:508: no cache hit)
        else:
            prompts = [wl.prompt() for _ in reqs]  # all prompts distinct: the speaker cache never hits (infer_v2.py:508)
        texts = [[torch.randint(2, 12000, (n,), generator=wl.g) for n in r] for r in reqs]
        audio_s = sum(audio_seconds([11 * n for n in r]) for r in reqs)
        step = lambda acc: run_mixed(wl, hp, reqs, prompts, texts, acc)
    else:
        pr = wl.prompt()
        texts = [torch.randint(2, 12000, (n_tok,), generator=wl.g) for _ in range(n_seg)]
        audio_s = R * audio_seconds([n_codes] * n_seg)
        step = lambda acc: run_request(wl, hp, pr, texts, n_codes, args.decode, R, acc)

    scratch = dict(wl.stage_ms)
    first_request_s = None
    for i in range(args.warmup):
        acc_w = dict(scratch)
        tw = time.perf_counter()
        step(acc_w)
        torch.cuda.synchronize()
        tw = time.perf_counter() - tw
        if i == 0:  # the first request of a fresh worker (graph capture, library / kernel-cache loads): reported, never timed
            first_request_s = tw
        log(f"warmup {i} done in {tw:.2f}s: " + ", ".join(f"{k} {v:.0f} ms" for k, v in acc_w.items()))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(wl.stage_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    total_audio = audio_s * args.steps
    if dist is not None:  # sum of the units over ranks, max of the times
        total_audio, elapsed = sharding.gather_throughput(total_audio, elapsed, device=dev)
    log(f"timed region: {elapsed:.2f}s for {args.steps} steps")
    ms_per_step = elapsed / args.steps * 1e3
    value = total_audio / elapsed
    stage_ms = wl.stage_ms

    # ---- roofline of the dominant kernel: the decode-step FC GEMV (largest weight stream per launch), timed live with
    # events on the launch stream, cycling the 24 layers (314 MB bf16 > Infinity Cache)
    roofline = None
    if rank == 0 and not args.no_roofline:
        B = min(n_seg, hp.gpt.max_batch) if not mixed and R == 1 and args.decode == "greedy" else hp.gpt.max_batch
        es = 2 if args.dtype == "bf16" else 4
        L = WR.GPT_CFG["layers"]
        alg_bytes = 4 * D * D * es + 4 * D * 4 + B * D * 4 + B * 4 * D * 4
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for l in range(L):
                hp.gpt.bench_gemv(2, l, B)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for l in range(L):
                    hp.gpt.bench_gemv(2, l, B)
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            e0.record()
            for _ in range(reps):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * L)
        achieved = alg_bytes / (us * 1e-6) / 1e9
        # HBM traffic per launch of the same kernel: NOT measured in this run -- read from the committed PMC passes of this
        # command (tools/pmc_traffic.py; separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 FETCH correction)
        traffic, traffic_src = None, None
        try:
            import glob

            wt = "__hip_bfloat16" if args.dtype == "bf16" else "float"
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[::-1]:
                ks = json.load(open(f))["kernels"]
                hit = [v for k, v in ks.items() if "gemv" in k and wt in k and f" {B}, " in k and v.get("role") == "fc"] or \
                      [v for k, v in ks.items() if f"gemv_reg_kernel<{wt}, 1280, 2, 2, {B}, 0, 2," in k]
                if hit:
                    traffic, traffic_src = round(hit[0]["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT)
                    break
        except Exception:
            traffic = None
        S_mid = P + max_codes // 2
        step_bytes = hp.gpt.step_bytes(B, S_mid)
        roofline = {
            "bound": "hbm", "kernel": f"decode LN2+FC GEMV ({args.dtype}, K=1280 -> 5120, gelu epilogue), B={B}",
            "achieved": round(achieved, 1), "peak": PEAK_HBM, "unit": "GB/s", "frac": round(achieved / PEAK_HBM, 4),
            "traffic": traffic, "traffic_source": (f"{traffic_src}: separate rocprofv3 --pmc passes of this command, not measured in this run" if traffic_src else None),
            "bytes_per_launch": alg_bytes, "us_per_launch": round(us, 3),
        }
        if not mixed and R == 1:
            step_us = stage_ms["gpt_gen"] / args.steps * 1e3 / n_codes / (n_seg if args.decode == "beam-turn" else 1)
            roofline["decode_step"] = {"alg_bytes": step_bytes, "us": round(step_us, 1), "achieved_GBps": round(step_bytes / step_us / 1e3, 1),
                                       "frac": round(step_bytes / step_us / 1e3 / PEAK_HBM, 4), "kernels_per_step": 5 * L + 2,
                                       "note": "gpt_gen / codes: includes the conditioning encoders, the prefills and the host syncs (pessimistic by ~3 %)"}

    # ---- the other stages against their rooflines (fp32 MFMA peak): BigVGAN from its timed stage, the DiT attention kernel
    # timed live with events on the launch stream (report the binding fraction per family)
    stage_roof = None
    if rank == 0 and not args.no_roofline and not mixed:
        bv_tf = hp.bigvgan.flops(1, frames) * n_seg * R / (stage_ms["bigvgan"] / args.steps * 1e-3) / 1e12
        stage_roof = {"bigvgan": dict(conv_roofline(bv_tf), flops_per_segment=hp.bigvgan.flops(1, frames))}
        if use_s2mel:
            from voice_tts_amd.s2mel import attn_full

            Ta = wl.Tref + frames
            qkv = torch.randn(2, Ta, 3, 8, 64, device=dev)
            for _ in range(3):
                attn_full(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                attn_full(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
            e1.record()
            torch.cuda.synchronize()
            us_a = e0.elapsed_time(e1) * 1e3 / 20
            fl = 4.0 * 2 * 8 * Ta * Ta * 64
            x3a = os.environ.get("IXTTS_ATTN_FULL", "x3") != "f32"
            tf_a, peak_a = fl / us_a / 1e6, (PEAK_BF16 / 6 if x3a else PEAK_F32)
            stage_roof["s2mel_attention"] = {"bound": "mfma", "kernel": ("attn_kv_planes + attn_full_x3_kernel + merge" if x3a else "attn_full_f32_kernel + merge") + " (B=2, H=8, T=%d)" % Ta,
                                             "achieved": round(tf_a, 1), "peak": round(peak_a, 1), "unit": "TFLOP/s", "frac": round(tf_a / peak_a, 3), "us_per_call": round(us_a, 1),
                                             "arithmetic": "fp32-equivalent flops; six bf16 MFMA partial products per fp32 product, fp32 accumulate" if x3a else "fp32 MFMA",
                                             "fp32_mfma_peak": PEAK_F32, "frac_of_fp32_mfma_peak": round(tf_a / PEAK_F32, 3)}

    # ---- extra figures of the default line (driver-visible): config 5, the served 3-beam default, fp32 parity mode, step per B
    extra = None
    default_line = world == 1 and not mixed and R == 1 and args.decode == "greedy" and args.dtype == "bf16" and not args.no_extra and args.emo_alpha is None
    if rank == 0 and default_line:
        extra = {}
        log("extra: BigVGAN config-5 microbench")
        mb = bigvgan_microbench(WR, dev, hp.bigvgan)
        extra["bigvgan_config5"] = {"ms_per_1000_frames": mb["value"], "audio_seconds_per_second": mb["audio_seconds_per_second"],
                                    "TFLOPs": mb["roofline"]["achieved"], "frac": mb["roofline"]["frac"], "peak": mb["roofline"]["peak"],
                                    "frac_fp32_mfma_peak": mb["roofline"]["frac_of_fp32_mfma_peak"]}
        one_audio = audio_seconds([n_codes] * n_seg)
        for name, dtype, mode, mb_ in (("beam3_bf16", "bf16", "beam", min(3 * n_seg, engine_max // 3 * 3)), ("beam3_bf16_segments_in_turn", "bf16", "beam-turn", 3),
                                       ("greedy_fp32", "f32", "greedy", min(max(2, n_seg), engine_max))):
            log(f"extra: {name} request")
            hp2 = make_hotpath(args, dev, dtype, mb_, P + n_codes + 64, frames, share=hp)
            if dtype == args.dtype:
                hp2.gpt.share_arena(hp.gpt)  # another engine shape over the same device weights
            else:
                hp2.gpt.load_state_dict(Wg)
            acc = {k: 0.0 for k in stage_ms}
            run_request(wl, hp2, pr, texts, n_codes, mode, 1, dict(acc))
            torch.cuda.synchronize()
            tq = time.perf_counter()
            reps = 2
            for _ in range(reps):
                run_request(wl, hp2, pr, texts, n_codes, mode, 1, acc)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - tq) / reps
            extra[name] = {"rtf": round(dt / one_audio, 5), "audio_seconds_per_second": round(one_audio / dt, 2), "ms_per_request": round(dt * 1e3, 1),
                           "stage_ms": {k: round(v / reps, 1) for k, v in acc.items()},
                           "decode_us_per_step": round(acc["gpt_gen"] / reps * 1e3 / n_codes / (n_seg if mode == "beam-turn" else 1), 1),
                           "sequences_per_step": {"beam": 3 * n_seg, "beam-turn": 3}.get(mode, n_seg)}
            del hp2
            torch.cuda.empty_cache()
        if use_s2mel and use_cond:
            log("extra: prompt-side stages at production size")
            try:
                wl.prompt_enc = build_prompt_encoder(hp, dev)
                wavs = [prompt_wav(5.0, 24000, seed=s_) for s_ in range(4)]
                wl.prompt_from_audio(wavs[0])
                torch.cuda.synchronize()
                tq = time.perf_counter()
                for w_ in wavs[1:]:
                    d_ = wl.prompt_from_audio(w_)
                torch.cuda.synchronize()
                extra["prompt_side"] = {"ms_per_new_prompt": round((time.perf_counter() - tq) / 3 * 1e3, 1), "prompt_seconds": 5.0,
                                        "frames": {"w2v_bert": int(d_["spk_cond_emb"].shape[1]), "ref_mel": int(d_["ref_mel"].shape[-1])},
                                        "note": "WAV bytes -> w2v-bert-2.0-shaped encoder (24 x 1024, random weights) -> RepCodec quantize -> reference mel -> "
                                                "kaldi fbank + CAM++ -> prompt condition; once per NEW speaker prompt (the reference caches it, infer_v2.py:508), "
                                                "outside the headline's timed region, inside mixed64's"}
                wl.prompt_enc = None
                torch.cuda.empty_cache()
            except Exception as e:  # (a transformers build without the model class)
                extra["prompt_side"] = {"error": str(e)[:200]}
        log("extra: decode step per batch size")
        # 1..4 sequences: the register GEMVs (what the judged line runs); 5..16: the wide engine (gpt_wide.h), also shown at 4
        extra["decode_step_by_batch"] = {}
        for kind, mb_, bs in (("register_gemv", min(4, engine_max), (1, 2, 4)), ("wide_mfma", engine_max, (4, 8, 16))):
            if kind == "wide_mfma" and engine_max <= 4:
                continue
            hpB = make_hotpath(args, dev, "bf16", mb_, P + 64 + 512, 64, share=hp)
            hpB.gpt.share_arena(hp.gpt)
            extra["decode_step_by_batch"][kind] = decode_step_by_batch(hpB, P, [b for b in bs if b <= mb_])
            del hpB
            torch.cuda.empty_cache()

    # ---- CPU baseline: the oracle (port of the reference's CPU path) on a bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not mixed:
        from oracle import gpt as OG
        from oracle import vocoder as OV

        torch.set_num_threads(host_cores())
        cores = torch.get_num_threads()
        log(f"cpu baseline on {cores} threads")
        orc = OG.GptOracle(Wg, WR.GPT_CFG["layers"], WR.GPT_CFG["heads"])
        g = wl.g
        cl0 = pr["conds"].cpu()
        fake, emb, mask = orc.prepare_gpt_inputs(cl0, texts[0])
        tc = time.perf_counter()
        logits, past = orc.prefill(emb, mask)
        t_prefill = time.perf_counter() - tc
        log(f"cpu prefill {t_prefill:.2f}s")
        n_dec = 64
        tc = time.perf_counter()
        tok = 5
        for k in range(1, n_dec + 1):
            logits, past = orc.decode_step(tok, k, past, mask)
            tok = int(torch.argmax(logits))
        t_step = (time.perf_counter() - tc) / n_dec
        log(f"cpu decode {t_step*1e3:.1f} ms/step")
        n_lat = 600
        tc = time.perf_counter()
        orc.latent_pass(cl0, texts[0], torch.randint(0, 8192, (n_lat,)))
        t_lat_row = (time.perf_counter() - tc) / (34 + n_tok + 2 + n_lat + 2)
        log(f"cpu latent {t_lat_row*1e3:.2f} ms/row")
        f_s = 256
        mel_s = (torch.randn(1, 80, f_s, generator=g) * 2 - 4).clamp(-11.5, 2)
        tc = time.perf_counter()
        OV.bigvgan_forward(mel_s, Wb)
        t_frame = (time.perf_counter() - tc) / f_s
        t_s2 = 0.0
        s2_note = ""
        if use_s2mel:
            # the s2mel glue is plain torch: its CPU leg is the same code on host tensors, 1 Euler step of the 25
            import voice_tts_amd.s2mel as S2

            cpu_s2 = S2.S2Mel(S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234), S2.S2MEL_CFG, device="cpu")
            lat_c = torch.randn(1, n_codes, D, generator=g)
            tc = time.perf_counter()
            cpu_s2(lat_c, torch.randint(0, 8192, (1, n_codes), generator=g), torch.tensor([n_codes]), pr["prompt_condition"].cpu(), pr["ref_mel"].cpu(),
                   pr["style"].cpu(), n_timesteps=1)
            t_s2 = (time.perf_counter() - tc) * 25
            s2_note = f", s2mel 1 of 25 Euler steps at T={wl.Tref + frames} ({t_s2 / 25:.2f} s/step)"
            log(f"cpu s2mel {t_s2 / 25:.2f} s/step")
        t_cond = 0.0
        cond_note = ""
        if use_cond:
            # same for the conditioning glue: the reference computes it per segment (infer_v2.py:629-635), so does this leg
            import voice_tts_amd.conditioning as CD

            cpu_cd = CD.Conditioning(CD.make_cond_weights(CD.COND_CFG, seed=1234), CD.COND_CFG, device="cpu")
            sc = pr["spk_cond_emb"].cpu()
            ls = torch.tensor([sc.shape[-1]])
            tc = time.perf_counter()
            with torch.no_grad():
                cpu_cd.merge_emovec(sc, sc, ls, ls, alpha=1.0)
                cpu_cd.get_conditioning(sc.transpose(1, 2), ls)
            t_cond = time.perf_counter() - tc
            cond_note = f", conditioning encoders on 249 frames ({t_cond:.2f} s per segment)"
            log(f"cpu conditioning {t_cond:.2f} s")
        est = n_seg * (t_cond + t_prefill + n_codes * t_step + (P + n_codes + 2) * t_lat_row + t_s2 + frames * t_frame)
        one_audio = audio_seconds([n_codes] * n_seg)
        cpu = {"value": round(one_audio / est, 4), "unit": "audio-s/s", "cores": cores, "kind": "port",
               "sample": f"oracle fp32: 1 prefill of {P} rows ({t_prefill:.2f}s), {n_dec} decode steps ({t_step*1e3:.1f} ms/step), "
                         f"latent pass on {n_lat} codes ({t_lat_row*1e3:.2f} ms/row), BigVGAN {f_s} frames ({t_frame*1e3:.1f} ms/frame){s2_note}{cond_note}; "
                         f"extrapolated linearly to the full request ({est:.0f}s est.)",
               "rtf": round(est / one_audio, 3)}

    if rank == 0:
        if mixed:
            n_segs = sum(len(r) for r in all_reqs)
            workload = (f"64 concurrent /tts requests, 50-400 characters (seed 5, {sum(sum(r) for r in all_reqs)} tokens in {n_segs} segments of <= 120 tokens), "
                        f"distinct 5 s prompts" + (" decoded and encoded inside the timed region (w2v-bert-2.0-shaped encoder, RepCodec, CAM++, mel: prompt.py)" if wl.prompt_enc is not None else " (synthetic per-prompt features)")
                        + f", request i -> rank i mod {world}; on each rank: conditioning encoders per request, all segments through the "
                        f"continuous-batching scheduler ({max_batch} decode slots, greedy fixed-length, 11 codes per token), then latent forward, "
                        + ("s2mel (25-step CFM), " if use_s2mel else "s2mel skipped, ") + "BigVGAN per segment; one step = the whole 64-request job")
        else:
            workload = ((f"1 /tts request per GPU: " if R == 1 else f"{R} concurrent /tts requests per GPU (segments share the decode slots, continuous batching B<={max_batch}), each ")
                        + f"{n_seg}x{n_tok}-token zh text segments (200-char utterance), "
                        + ("conditioning encoders on 249 prompt frames (conformer + perceiver, PyTorch-ROCm glue, fp32), " if use_cond else "")
                        + (f"emotion prompt merged at alpha {args.emo_alpha}, " if args.emo_alpha is not None else "")
                        + (f"greedy fixed-length decode {n_codes} codes/segment batched B={min(n_seg, max_batch)}, " if args.decode == "greedy" else
                           f"top-p sampling (top_p 0.8, top_k 30, T 0.8, theta 10) fixed-length decode {n_codes} codes/segment batched B={min(n_seg, max_batch)}, " if args.decode == "sample" else
                           f"3-beam beam-sample (top_k 30, top_p 0.8, T 0.8, theta 10) fixed-length decode {n_codes} codes/segment, "
                           + ("the segments' beam groups stepping together (wide engine), " if args.decode == "beam" else "segments in turn, "))
                        + "latent GPT forward, "
                        + ("s2mel (length regulator + 25-step CFM/DiT, PyTorch-ROCm glue, fp32, 430-frame prompt), " if use_s2mel else "s2mel skipped (synthetic mel), ")
                        + f"BigVGAN {frames} frames/segment -> {audio_s / R:.2f} s audio; the per-prompt features the reference caches (w2v-bert, CAM++, "
                        f"semantic codec, reference mel; built in voice-tts_amd/prompt.py) enter as synthetic HBM-resident tensors of the production shapes"
                        + ("" if use_cond else " and so does conds_latent"))
        out = {
            "metric": "audio_seconds_per_second", "value": round(value, 3), "unit": "audio-s/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "strong" if mixed else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            # weak: wall seconds per audio second of ONE request stream; mixed64: of the whole job
            "rtf": round(elapsed / total_audio * (1 if mixed else world), 5),
            "config": {
                "workload": workload,
                "segments": n_seg if not mixed else n_segs, "text_tokens_per_segment": n_tok if not mixed else "50-400 chars / ceil(len/120)",
                "codes_per_segment": n_codes if not mixed else "11 per token", "mel_frames_per_segment": frames if not mixed else "floor(1.72 codes)",
                "audio_seconds_per_step": round(total_audio / args.steps, 3), "parallelism": f"request-per-GPU x{world}, RCCL weight broadcast at load",
                "gpt_precision": f"{args.dtype} weights+KV, fp32 accumulate", "bigvgan_precision": "fp32 tensors; conv products as six bf16 MFMA partial products of exactly split operands, fp32 accumulate" if os.environ.get("IXTTS_BV_CONV", "x3") != "f32" else "fp32 (fp32 MFMA)",
                "s2mel": "torch fp32 glue (library fp32 GEMMs) + HIP attention (split-product bf16 MFMA, fp32-accurate) / row kernels in the timed region" if use_s2mel else "excluded",
                "conditioning": "torch fp32 glue in the timed region (inside gpt_gen)" if use_cond else "excluded",
            },
            "stage_ms_per_step": {k: round(v / args.steps, 2) for k, v in stage_ms.items()},
            "load_s": round(t_load, 1),
            "load_warmup_s": round(t_wu, 2),
            "first_request_s": None if first_request_s is None else round(first_request_s, 2),
            "roofline": roofline,
            "stage_rooflines": stage_roof,
            "extra": extra,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
