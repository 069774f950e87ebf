"""Per-kernel view of one wide decode configuration for rocprofv3 --kernel-trace --stats: python tools/wide_prof.py B [P] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import voice_tts_amd.weights as WR
from voice_tts_amd.gpt_engine import GptEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = int(sys.argv[2]) if len(sys.argv) > 2 else 137
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
emb = (torch.randn(P - 1, 1280, generator=torch.Generator().manual_seed(1)) * 0.5).to(dev)
eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=P + N + 96, max_batch=max(B, 5), device=dev).load_state_dict(W)
for b in range(B):
    eng.prefill(b, emb, 0)
eng.decode(B, N, repetition_penalty=10.0, suppress_stop=True)
torch.cuda.synchronize()
