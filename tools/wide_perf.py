"""Decode step per batch size on the register GEMVs (max_batch 4) and on the wide MFMA engine (max_batch 16): us per step,
us per sequence, fraction of the HBM roofline.  Run on the GPU box: python tools/wide_perf.py [P] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import voice_tts_amd.weights as WR
from voice_tts_amd.gpt_engine import GptEngine

P = int(sys.argv[1]) if len(sys.argv) > 1 else 137
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
emb = (torch.randn(P - 1, 1280, generator=torch.Generator().manual_seed(1)) * 0.5).to(dev)
out = {}
for name, mb, batches in (("register", 4, (1, 2, 4)), ("wide", 16, (1, 2, 4, 8, 12, 16))):
    eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=P + N + 96, max_batch=mb, device=dev).load_state_dict(W)
    for B in batches:
        for b in range(B):
            eng.prefill(b, emb, 0)
        eng.decode(B, 16, repetition_penalty=10.0, suppress_stop=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.decode(B, N, repetition_penalty=10.0, suppress_stop=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / N
        by = eng.step_bytes(B, P + 16 + N // 2)
        out[f"{name}_B{B}"] = dict(us=round(us, 1), us_per_seq=round(us / B, 1), frac=round(by / us / 1e3 / 8000, 4))
        print(name, B, out[f"{name}_B{B}"], flush=True)
    del eng
    torch.cuda.empty_cache()
print(json.dumps(out))
