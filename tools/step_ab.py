"""A/B of decode-chain levers: step time per batch on whatever library IXTTS_LIB names (variant builds: IXTTS_VARIANT=name
IXTTS_EXP="-D..." python -m voice_tts_amd.build), bf16, 1100 greedy steps from a 137-row prompt (the bench's contexts), plus a
checksum of the tokens (the levers must not change a single one)."""
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
tag = os.path.basename(os.environ.get("IXTTS_LIB", "default")) + " PF_KIB=" + os.environ.get("IXTTS_PF_KIB", "-")
owner = None
for B in [int(x) for x in (sys.argv[1:] or ["1", "2", "3"])]:
    eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=137 + 1100 + 64, max_batch=B, device=dev)
    if owner is None:
        owner = eng.load_state_dict(W)
    else:
        eng.share_arena(owner)
    best = 1e9
    for rep in range(3):
        for b in range(B):
            eng.prefill(b, emb[: 136 - 7 * b], 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode(B, 1100, suppress_stop=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 1100 * 1e6)
    h = hashlib.sha1(b"".join(eng.read(b)[0].tobytes() for b in range(B))).hexdigest()[:10]
    print(f"[{tag}] B={B}: {best:.1f} us/step  tokens {h}", flush=True)
