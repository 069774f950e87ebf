"""A/B on the wide engine (IXTTS_LIB): step time at B sequences, bf16, 1100 greedy steps from a 137-row prompt + token checksum."""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_tts_amd.weights as WR  # noqa: E402
from voice_tts_amd.gpt_engine import GptEngine  # noqa: E402

dev = torch.device("cuda:0")
W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
emb = torch.randn(136, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
tag = os.path.basename(os.environ.get("IXTTS_LIB", "default")) + " attn=" + os.environ.get("IXTTS_WIDE_ATTN", "split")
eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=137 + 1100 + 64, max_batch=16, device=dev).load_state_dict(W)
for B in [int(x) for x in (sys.argv[1:] or ["6", "8", "16"])]:
    best = 1e9
    for rep in range(3):
        for b in range(B):
            eng.prefill(b, emb[: 136 - 3 * b], 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode(B, 1100, suppress_stop=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 1100 * 1e6)
    h = hashlib.sha1(b"".join(eng.read(b)[0].tobytes() for b in range(B))).hexdigest()[:10]
    print(f"[{tag}] wide B={B}: {best:.1f} us/step  tokens {h}", flush=True)
