#!/usr/bin/env python
"""bench.py -- throughput of the MI355X hot path on BASELINE.json's metric.

Workload (configs[1], SURVEY.md 8(d) "Config 2"): ONE /tts request = 200-token zh text ->
2 segments x 100 tokens (prompt P = 137 rows), 5 s speaker prompt already reduced to
`conds_latent`; greedy fixed-length decode n = 1100 codes per segment (stop token
suppressed: random weights never emit EOS), the latent GPT forward per segment, and
BigVGAN over floor(1.72 n) = 1892 mel frames per segment -> 44.13 s of audio.
The stages `north_star` leaves to PyTorch glue (conditioning encoders, s2mel CFM) are not
built in this repo and are NOT in the timed region: their outputs (`conds_latent`, mel)
are synthetic tensors resident in HBM, so `value` is the hot-path-only rate.

One "step" = one such request.  Weak scaling: every rank serves its own request
(process-per-GPU sharding, gunicorn_config.py:43-60); RCCL is used only to broadcast the
packed weight arenas from rank 0 at load.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch


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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--tokens", type=int, default=100, help="text tokens per segment")
    ap.add_argument("--codes", type=int, default=1100, help="mel codes per segment (11 per char)")
    ap.add_argument("--segments", type=int, default=2)
    ap.add_argument("--no-s2mel", action="store_true", help="leave the PyTorch-glue s2mel stage out of the timed region (feed synthetic mels)")
    ap.add_argument("--concurrency", type=int, default=1, help="requests in flight per GPU (config 3 style): their segments share the decode "
                    "slots through the continuous-batching scheduler (row N3); 1 = BASELINE config[1], the judged line")
    ap.add_argument("--no-cond", action="store_true", help="leave the conditioning encoders out of the timed region (feed a synthetic conds_latent)")
    ap.add_argument("--decode", default="greedy", choices=["greedy", "beam"], help="greedy: BASELINE config[1] (the judged line); beam: the served "
                    "default of config[2] -- 3-beam beam-sample, top_k 30, top_p 0.8, temperature 0.8 -- one segment at a time")
    ap.add_argument("--bigvgan-only", action="store_true", help="BASELINE config[4]: 1000-frame random mel -> waveform microbench (20 warm-up + 100 timed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def bigvgan_microbench(WR, dev, rank):
    """BASELINE config[4] / SURVEY 8(d) config 5: mel ~ N(-4, 2) clipped to [-11.5, 2], fp32 [1, 80, 1000] -> [1, 1, 256000]."""
    from voice_tts_amd.bigvgan import BigVGAN

    m = BigVGAN(WR.BIGVGAN_CFG, max_frames=1024, device=dev).load_state_dict(WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234))
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
    if rank == 0:
        print(json.dumps({"metric": "bigvgan_ms_per_1000_frames", "value": round(ms, 3), "unit": "ms", "n_gpus": 1, "steps": 100, "warmup": 20,
                          "higher_is_better": False, "dtype": "f32", "data": "synthetic", "audio_seconds_per_second": round(256 * F / 22050 / (ms * 1e-3), 1),
                          "config": {"workload": "BigVGAN-only: 1000-frame random mel -> 256000 samples (11.61 s)"},
                          "roofline": {"bound": "mfma", "achieved": round(fl / ms / 1e9, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / 157.3, 3),
                                       "flops": fl, "hbm_GBps_algorithmic": round(alg_bytes / (ms * 1e-3) / 1e9, 1), "hbm_peak_GBps": 8000.0}}), flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    import voice_tts_amd.weights as WR
    from voice_tts_amd.pipeline import HotPath, audio_seconds

    if args.bigvgan_only:
        return bigvgan_microbench(WR, dev, rank)
    D = WR.GPT_CFG["model_dim"]
    n_seg, n_tok, n_codes = args.segments, args.tokens, args.codes
    frames = int(n_codes * 1.72)
    P = 34 + n_tok + 2 + 1
    R = max(1, args.concurrency)
    hp = HotPath(dtype=args.dtype, device=dev, max_batch=4 if R > 1 else (3 if args.decode == "beam" else (max(2, n_seg) if n_seg <= 4 else 4)), max_seq=P + n_codes + 64,
                 max_frames=frames)

    # ---- load: rank 0 builds the (synthetic, seeded) weights, RCCL broadcasts the packed arenas
    t_load = time.time()
    Wg = Wb = None
    use_s2mel = not args.no_s2mel
    if rank == 0:
        Wg = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
        Wb = WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234)
        hp.load(Wg, Wb)
    if use_s2mel:
        # PyTorch glue stage (row N1): every rank regenerates the same seeded weights (98 M params, plain torch tensors)
        import voice_tts_amd.s2mel as S2

        hp.attach_s2mel(S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234))
    use_cond = not args.no_cond
    if use_cond:
        # PyTorch glue stage (row N2): conformer + perceiver conditioning encoders (338 M params, seeded per rank alike)
        import voice_tts_amd.conditioning as CD

        hp.attach_conditioning(CD.make_cond_weights(CD.COND_CFG, seed=1234))
    if world > 1:
        for t in hp.broadcast_tensors():
            dist.broadcast(t, src=0)
        if rank != 0:
            hp.adopt()
    torch.cuda.synchronize()
    t_load = time.time() - t_load
    log(f"weights loaded in {t_load:.1f}s")

    # ---- synthetic request, resident in HBM before the timed region (seeded per rank)
    g = torch.Generator().manual_seed(100 + rank)
    conds = [(torch.randn(34, D, generator=g) * 0.5).to(dev) for _ in range(n_seg)]
    # 5 s speaker prompt -> 249 w2v-bert frames: stand-in for the cached, normalised layer-17 features
    spk_cond_emb = torch.randn(1, 249, 1024, generator=g).to(dev)
    texts = [torch.randint(2, 12000, (n_tok,), generator=g) for _ in range(n_seg)]
    mels = [(torch.randn(1, 80, frames, generator=g) * 2 - 4).clamp(-11.5, 2).to(dev) for _ in range(n_seg)]
    # 5 s speaker prompt -> 430 reference mel frames: stand-ins for the cached prompt features
    Tref = 430
    prompt_condition = torch.randn(1, Tref, 512, generator=g).to(dev)
    ref_mel = (torch.randn(1, 80, Tref, generator=g) * 2 - 4).clamp(-11.5, 2).to(dev)
    style = torch.randn(1, 192, generator=g).to(dev)
    audio_s = audio_seconds([n_codes] * n_seg)

    stage_ms = {"gpt_gen": 0.0, "gpt_forward": 0.0, "s2mel": 0.0, "bigvgan": 0.0}

    def request(timed):
        def tick():
            torch.cuda.synchronize()
            return time.perf_counter()

        t0 = tick()
        if use_cond:  # merge_emovec + get_conditioning (infer_v2.py:629-635, model_v2.py:684-696), once per request
            cl = hp.conds_from_prompt(spk_cond_emb)
            # synthetic encoder weights give arbitrary latent statistics; keep the GPT prefix at the scale it is built for
            cl = cl * (0.5 / cl.std().clamp_min(1e-6))
            for s in range(n_seg):
                conds[s] = cl
        prompts = [hp.prepare_gpt_inputs(conds[s], texts[s])[:2] for s in range(n_seg)]
        if R > 1:  # R requests in flight: all their segments go through the continuous-batching scheduler
            many = hp.generate_many([(e, p, n_codes) for _ in range(R) for (e, p) in prompts], fixed_length=True, repetition_penalty=10.0)
            codes = many[:n_seg]
        elif args.decode == "beam":  # served default (infer_v2.py:598-606): beams of one segment occupy the slots, segments in turn
            codes = []
            for e, p in prompts:
                hp.gpt.prefill(0, e, p)
                hp.gpt.beam_begin(3)
                hp.gpt.beam_decode(n_codes, repetition_penalty=10.0, temperature=0.8, top_k=30, top_p=0.8, suppress_stop=True, seed=7)
                codes.append(hp.gpt.beam_read(n_codes)[0][:n_codes])
        else:
            codes = hp.generate(prompts, n_codes, repetition_penalty=10.0, fixed_length=True)
        t1 = tick()
        t2 = t2b = t3 = t1
        for r in range(R):  # the post-decode stages stay per segment (and per request)
            ta = tick()
            rc = codes if R == 1 else many[r * n_seg:(r + 1) * n_seg]
            lats = [hp.latent(conds[s], texts[s], rc[s]) for s in range(n_seg)]
            tb = tick()
            if use_s2mel:  # 25 Euler steps x CFG batch 2 over T = 430 + 1892 frames, fp32 (infer_v2.py:713-731)
                seg_mels = [hp.s2mel(lats[s], rc[s], prompt_condition, ref_mel, style) for s in range(n_seg)]
                # synthetic weights give arbitrary mel statistics; keep the vocoder input in the log-mel range it is built for
                seg_mels = [m.clamp(-11.5, 2.0) for m in seg_mels]
            else:
                seg_mels = mels
            tc_ = tick()
            wavs = []
            for s in range(n_seg):
                w = hp.vocode(seg_mels[s])
                wavs.append(w.to(torch.int16).cpu())  # wav.cpu() per segment, int16 truncation (infer_v2.py:744,781)
            td = tick()
            t2, t2b, t3 = t2 + (tb - ta), t2b + (tc_ - ta), t3 + (td - ta)
        if timed:
            stage_ms["gpt_gen"] += (t1 - t0) * 1e3
            stage_ms["gpt_forward"] += (t2 - t1) * 1e3
            stage_ms["s2mel"] += (t2b - t2) * 1e3
            stage_ms["bigvgan"] += (t3 - t2b) * 1e3
        assert all(len(c) == n_codes for c in codes) and all(l.shape == (n_codes, D) for l in lats)
        assert all(w.shape[-1] == frames * 256 for w in wavs)
        return codes

    for i in range(args.warmup):
        request(False)
        log(f"warmup {i} done")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        request(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    log(f"timed region: {elapsed:.2f}s for {args.steps} steps")
    ms_per_step = elapsed / args.steps * 1e3
    value = world * R * audio_s * args.steps / elapsed

    # ---- roofline of the dominant kernel: the decode-step FC GEMV (largest weight stream per launch),
    # timed live with events on the launch stream, cycling the 24 layers (314 MB bf16 > Infinity Cache)
    roofline = None
    if rank == 0 and not args.no_roofline:
        B = n_seg if n_seg <= hp.gpt.max_batch else 1
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
        # HBM traffic per launch of the same kernel from the committed PMC passes (tools/pmc_traffic.py; separate
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 FETCH correction) -- null if no pass matches
        traffic = None
        try:
            import glob

            wt = "__hip_bfloat16" if args.dtype == "bf16" else "float"
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[::-1]:
                ks = json.load(open(f))["kernels"]
                hit = [v for k, v in ks.items() if f"gemv_reg_kernel<{wt}, 1280, 2, 2, {B}, 0, 2," in k]
                if hit:
                    traffic = round(hit[0]["hbm_bytes_per_launch"])
                    break
        except Exception:
            traffic = None
        S_mid = P + n_codes // 2
        step_bytes = hp.gpt.step_bytes(B, S_mid)
        step_us = stage_ms["gpt_gen"] / args.steps * 1e3 / n_codes  # includes the conditioning encoders, both prefills and the host syncs (pessimistic by ~3 %)
        roofline = {
            "bound": "hbm", "kernel": f"gemv_reg_kernel<{args.dtype},K=1280,IN_LN,EPI_GELU> (decode LN2+FC, B={B})",
            "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
            "traffic": traffic, "bytes_per_launch": alg_bytes, "us_per_launch": round(us, 3),
            "decode_step": {"alg_bytes": step_bytes, "us": round(step_us, 1), "achieved_GBps": round(step_bytes / step_us / 1e3, 1),
                            "frac": round(step_bytes / step_us / 1e3 / 8000.0, 4), "kernels_per_step": 5 * L + 2},
        }

    # ---- the other stages against their rooflines (fp32 MFMA peak 157.3 TFLOP/s): BigVGAN from its timed stage, the DiT
    # attention kernel timed live with events on the launch stream (report the binding fraction per family)
    stage_roof = None
    if rank == 0 and not args.no_roofline:
        PEAK_F32 = 157.3
        bv_tf = hp.bigvgan.flops(1, frames) * n_seg * R / (stage_ms["bigvgan"] / args.steps * 1e-3) / 1e12
        stage_roof = {"bigvgan": {"bound": "mfma", "achieved": round(bv_tf, 1), "peak": PEAK_F32, "unit": "TFLOP/s", "frac": round(bv_tf / PEAK_F32, 3),
                                  "flops_per_segment": hp.bigvgan.flops(1, frames)}}
        if use_s2mel:
            from voice_tts_amd.s2mel import attn_full

            Ta = Tref + frames
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
            stage_roof["s2mel_attention"] = {"bound": "mfma", "kernel": "attn_full_f32_kernel + merge (B=2, H=8, T=%d)" % Ta, "achieved": round(fl / us_a / 1e6, 1),
                                             "peak": PEAK_F32, "unit": "TFLOP/s", "frac": round(fl / us_a / 1e6 / PEAK_F32, 3), "us_per_call": round(us_a, 1)}

    # ---- CPU baseline: the oracle (port of the reference's CPU path) on a bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import gpt as OG
        from oracle import vocoder as OV

        torch.set_num_threads(host_cores())
        cores = torch.get_num_threads()
        log(f"cpu baseline on {cores} threads")
        orc = OG.GptOracle(Wg, WR.GPT_CFG["layers"], WR.GPT_CFG["heads"])
        fake, emb, mask = orc.prepare_gpt_inputs(conds[0].cpu(), texts[0])
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
        orc.latent_pass(conds[0].cpu(), texts[0], torch.randint(0, 8192, (n_lat,)))
        t_lat_row = (time.perf_counter() - tc) / (34 + n_tok + 2 + n_lat + 2)
        log(f"cpu latent {t_lat_row*1e3:.2f} ms/row")
        f_s = 256
        tc = time.perf_counter()
        OV.bigvgan_forward(mels[0][:, :, :f_s].cpu(), Wb)
        t_frame = (time.perf_counter() - tc) / f_s
        t_s2 = 0.0
        s2_note = ""
        if use_s2mel:
            # the s2mel glue is plain torch: its CPU leg is the same code on host tensors, 1 Euler step of the 25
            import voice_tts_amd.s2mel as S2

            cpu_s2 = S2.S2Mel(S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234), S2.S2MEL_CFG, device="cpu")
            lat_c = torch.randn(1, n_codes, D, generator=g)
            tc = time.perf_counter()
            cpu_s2(lat_c, torch.randint(0, 8192, (1, n_codes), generator=g), torch.tensor([n_codes]), prompt_condition.cpu(), ref_mel.cpu(),
                   style.cpu(), n_timesteps=1)
            t_s2 = (time.perf_counter() - tc) * 25
            s2_note = f", s2mel 1 of 25 Euler steps at T={Tref + frames} ({t_s2 / 25:.2f} s/step)"
            log(f"cpu s2mel {t_s2 / 25:.2f} s/step")
        t_cond = 0.0
        cond_note = ""
        if use_cond:
            # same for the conditioning glue: the reference computes it per segment (infer_v2.py:629-635), so does this leg
            import voice_tts_amd.conditioning as CD

            cpu_cd = CD.Conditioning(CD.make_cond_weights(CD.COND_CFG, seed=1234), CD.COND_CFG, device="cpu")
            sc = spk_cond_emb.cpu()
            ls = torch.tensor([sc.shape[-1]])
            tc = time.perf_counter()
            with torch.no_grad():
                cpu_cd.merge_emovec(sc, sc, ls, ls, alpha=1.0)
                cpu_cd.get_conditioning(sc.transpose(1, 2), ls)
            t_cond = time.perf_counter() - tc
            cond_note = f", conditioning encoders on 249 frames ({t_cond:.2f} s per segment)"
            log(f"cpu conditioning {t_cond:.2f} s")
        est = n_seg * (t_cond + t_prefill + n_codes * t_step + (P + n_codes + 2) * t_lat_row + t_s2 + frames * t_frame)
        cpu = {"value": round(audio_s / est, 4), "unit": "audio-s/s", "cores": cores, "kind": "port",
               "sample": f"oracle fp32: 1 prefill of {P} rows ({t_prefill:.2f}s), {n_dec} decode steps ({t_step*1e3:.1f} ms/step), "
                         f"latent pass on {n_lat} codes ({t_lat_row*1e3:.2f} ms/row), BigVGAN {f_s} frames ({t_frame*1e3:.1f} ms/frame){s2_note}{cond_note}; "
                         f"extrapolated linearly to the full request ({est:.0f}s est.)",
               "rtf": round(est / audio_s, 3)}

    if rank == 0:
        out = {
            "metric": "audio_seconds_per_second", "value": round(value, 3), "unit": "audio-s/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "rtf": round(elapsed / (R * audio_s * args.steps), 5),
            "config": {
                "workload": (f"1 /tts request per GPU: " if R == 1 else f"{R} concurrent /tts requests per GPU (segments share the decode slots, continuous batching B<=4), each ")
                            + f"{n_seg}x{n_tok}-token zh text segments (200-char utterance), "
                            + ("conditioning encoders on 249 prompt frames (conformer + perceiver, PyTorch-ROCm glue, fp32), " if use_cond else "")
                            + (f"greedy fixed-length decode {n_codes} codes/segment batched B={n_seg}, " if args.decode == "greedy" else
                               f"3-beam beam-sample (top_k 30, top_p 0.8, T 0.8, theta 10) fixed-length decode {n_codes} codes/segment, segments in turn, ")
                            + "latent GPT forward, "
                            + ("s2mel (length regulator + 25-step CFM/DiT, PyTorch-ROCm glue, fp32, 430-frame prompt), " if use_s2mel else "s2mel skipped (synthetic mel), ")
                            + f"BigVGAN {frames} frames/segment -> {audio_s:.2f} s audio; prompt feature extraction (w2v-bert, CAM++, semantic "
                            f"codec, reference mel) not built: spk_cond_emb / prompt_condition / ref_mel / style are synthetic HBM-resident inputs"
                            + ("" if use_cond else " and so is conds_latent"),
                "segments": n_seg, "text_tokens_per_segment": n_tok, "codes_per_segment": n_codes, "mel_frames_per_segment": frames,
                "audio_seconds_per_request": round(audio_s, 3), "parallelism": f"request-per-GPU x{world}, RCCL weight broadcast at load",
                "gpt_precision": f"{args.dtype} weights+KV, fp32 accumulate", "bigvgan_precision": "fp32 (fp32 MFMA)",
                "s2mel": "torch fp32 glue + HIP attention / row kernels in the timed region" if use_s2mel else "excluded",
                "conditioning": "torch fp32 glue in the timed region (inside gpt_gen)" if use_cond else "excluded",
            },
            "stage_ms_per_step": {k: round(v / args.steps, 2) for k, v in stage_ms.items()},
            "load_s": round(t_load, 1),
            "roofline": roofline,
            "stage_rooflines": stage_roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
