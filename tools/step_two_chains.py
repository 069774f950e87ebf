"""Two segments as two INDEPENDENT one-sequence chains on two streams (two engines over one weight arena) against one two-sequence
chain: does the latency-bound step hide a second chain?  bf16, 1100 greedy steps from 137 / 130 rows, tokens must be identical."""
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
N = 1100


def mk(B, owner=None):
    e = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=137 + N + 64, max_batch=B, device=dev)
    if owner is None:
        e.load_state_dict(W)
    else:
        e.share_arena(owner)
    return e


def sha(parts):
    return hashlib.sha1(b"".join(p.tobytes() for p in parts)).hexdigest()[:10]


e2 = mk(2)
ea, eb = mk(1, e2), mk(1, e2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for rep in range(3):
    for b in range(2):
        e2.prefill(b, emb[: 136 - 7 * b], 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e2.decode(2, N, suppress_stop=True)
    torch.cuda.synchronize()
    t_one = (time.perf_counter() - t0) / N * 1e6
    ea.prefill(0, emb[:136], 0)
    eb.prefill(0, emb[:129], 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ea.decode(1, N, suppress_stop=True)
    torch.cuda.synchronize()
    t_b1 = (time.perf_counter() - t0) / N * 1e6
    ea.prefill(0, emb[:136], 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        ea.decode(1, N, suppress_stop=True)
    with torch.cuda.stream(sb):
        eb.decode(1, N, suppress_stop=True)
    torch.cuda.synchronize()
    t_two = (time.perf_counter() - t0) / N * 1e6
    same = sha([e2.read(0)[0], e2.read(1)[0]]) == sha([ea.read(0)[0], eb.read(0)[0]])
    print(f"rep {rep}: one B=2 chain {t_one:.1f} us/step | one B=1 chain {t_b1:.1f} | two B=1 chains on two streams {t_two:.1f}  tokens same: {same}", flush=True)
