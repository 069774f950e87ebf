"""Spike: does data a kernel read stay in the XCD L2s across a kernel boundary?  FC GEMV of layer l launched
1x / 2x / 4x in a row per layer (same grid -> same WG->XCD mapping), cycling all layers; us per launch."""
import sys
import torch
sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
from voice_tts_amd.gpt_engine import GptEngine

dev = torch.device("cuda:0")
W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
B = 2
eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=2048, max_batch=B, device=dev).load_state_dict(W)
emb = torch.randn(136, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
for b in range(B):
    eng.prefill(b, emb, 0)
L = WR.GPT_CFG["layers"]
for which, name in ((2, "fc"), (1, "out-proj"), (3, "mlp-proj"), (0, "qkv")):
    for rep in (1, 2, 4):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for l in range(L):
                eng.bench_gemv(which, l, B)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for l in range(L):
                    for _ in range(rep):
                        eng.bench_gemv(which, l, B)
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (20 * L * rep)
        print(f"{name}: {rep}x in a row -> {us:.2f} us/launch", flush=True)
