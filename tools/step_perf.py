"""Decode step time per batch (register GEMVs), greedy, bf16, context ~300-700: us/step."""
import sys, time, torch
sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
from voice_tts_amd.gpt_engine import GptEngine
dev = torch.device("cuda:0")
W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
emb = torch.randn(136, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
for B in [int(x) for x in (sys.argv[1:] or ["2", "3"])]:
    eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=2048, max_batch=B, device=dev).load_state_dict(W)
    for b in range(B):
        eng.prefill(b, emb, 0)
    eng.decode(B, 64, suppress_stop=True)
    torch.cuda.synchronize(); t0 = time.time()
    eng.decode(B, 800, suppress_stop=True)
    torch.cuda.synchronize()
    print(f"B={B}: {(time.time()-t0)/800*1e6:.1f} us/step", flush=True)
    del eng
