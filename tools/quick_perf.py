"""Developer timing probe (not the judged bench): BigVGAN F=1000 and GPT decode steps."""
import sys
import time

import torch

sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
from voice_tts_amd.bigvgan import BigVGAN
from voice_tts_amd.gpt_engine import GptEngine

dev = torch.device("cuda:0")
what = sys.argv[1:] or ["bigvgan", "gpt"]


def ev_time(fn, n):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


if "bigvgan" in what:
    t0 = time.time()
    W = WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234)
    m = BigVGAN(WR.BIGVGAN_CFG, max_frames=2048, fast_sin="fast" in what, device=dev).load_state_dict(W)
    print(f"bigvgan load {time.time()-t0:.1f}s", flush=True)
    for F in (1000, 1892):
        mel = (torch.randn(1, 80, F, generator=torch.Generator().manual_seed(6)) * 2 - 4).clamp(-11.5, 2).to(dev)
        for _ in range(2):
            w = m(mel)
        ms = ev_time(lambda: m(mel), 5)
        fl = m.flops(1, F)
        print(f"bigvgan F={F}: {ms:.2f} ms  {fl/ms/1e9:.1f} TFLOP/s  audio {256*F/22050:.2f}s  absmax {w.abs().max().item():.3f} clipped {(w.abs()>=1).float().mean().item():.4f}", flush=True)
    del m

if "gpt" in what:
    t0 = time.time()
    W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
    print(f"gpt weights {time.time()-t0:.1f}s", flush=True)
    for dtype in (("bf16",) if "bf16only" in what else ("bf16", "f32")):
        for B in (1, 2):
            t0 = time.time()
            eng = GptEngine(WR.GPT_CFG, dtype=dtype, max_seq=2048, max_batch=B, device=dev).load_state_dict(W)
            tl = time.time() - t0
            emb = torch.randn(136, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
            t0 = time.time()
            for b in range(B):
                eng.prefill(b, emb, 0)
            torch.cuda.synchronize()
            tp = time.time() - t0
            eng.decode(B, 20, suppress_stop=True)
            ms = ev_time(lambda: eng.decode(B, 100, suppress_stop=True), 3) / 100
            S = 137 + 20 + 150
            by = eng.step_bytes(B, S)
            print(f"gpt {dtype} B={B}: load {tl:.1f}s prefill {tp*1e3:.0f} ms  step {ms*1e3:.1f} us  {by/ms/1e6:.0f} GB/s alg  tok/s {B/ms*1e3:.0f}", flush=True)
            ids, fin = eng.read(0)
            print("   ids", ids[:8].tolist(), len(ids))
            del eng

if "micro" in what:
    # per-kernel cost inside a graph, un-profiled: 24 layers x kind, replayed
    W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
    for dtype in (("bf16",) if "bf16only" in what else ("bf16", "f32")):
        eng = GptEngine(WR.GPT_CFG, dtype=dtype, max_seq=2048, max_batch=2, device=dev).load_state_dict(W)
        emb = torch.randn(136, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
        eng.prefill(0, emb, 0)
        eng.prefill(1, emb, 0)
        es = 2 if dtype == "bf16" else 4
        sizes = {0: 3 * 1280 * 1280 * es, 1: 1280 * 1280 * es, 2: 4 * 1280 * 1280 * es, 3: 4 * 1280 * 1280 * es, 4: 8194 * 1280 * es}
        names = {0: "qkv", 1: "out", 2: "fc", 3: "mlp_out", 4: "head"}
        s = torch.cuda.Stream()
        for B in (1, 2):
            for which in range(5):
                with torch.cuda.stream(s):
                    for l in range(24):
                        eng.bench_gemv(which, l, B)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        for l in range(24):
                            eng.bench_gemv(which, l if which < 4 else 0, B)
                    g.replay()
                    ms = ev_time(lambda: g.replay(), 20) / 24
                print(f"micro {dtype} B={B} {names[which]:8s}: {ms*1e3:6.2f} us/kernel  {sizes[which]/ms/1e6:7.0f} GB/s", flush=True)
        del eng
    x = torch.zeros(64, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x.add_(1)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(200):
                x.add_(1)
        g.replay()
        ms = ev_time(lambda: g.replay(), 20) / 200
    print(f"micro trivial kernel in graph: {ms*1e3:.2f} us", flush=True)

if "beam" in what:
    W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
    eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=2048, max_batch=3, device=dev).load_state_dict(W)
    emb = torch.randn(136, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
    eng.prefill(0, emb, 0)
    eng.beam_begin(3)
    eng.beam_decode(16, suppress_stop=True)
    ms = ev_time(lambda: eng.beam_decode(96, suppress_stop=True), 3) / 96
    ids, done, score, bs, lt, src = eng.beam_read(2000)
    print(f"beam-sample bf16 3 beams: {ms*1e3:.1f} us/step, {len(ids)} ids, score {score:.1f}, done {done}", flush=True)
