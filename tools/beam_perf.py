"""Served-default decode (3-beam beam-sample, top_k 30, top_p 0.8, T 0.8, theta 10): us per step."""
import sys, time, torch
sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
from voice_tts_amd.gpt_engine import GptEngine
dev = torch.device("cuda:0")
eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=2048, max_batch=3, device=dev).load_state_dict(WR.make_gpt_weights(WR.GPT_CFG, seed=1234))
emb = torch.randn(136, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
for n in (64, 1000):
    eng.prefill(0, emb, 0)
    eng.beam_begin(3)
    torch.cuda.synchronize(); t0 = time.time()
    eng.beam_decode(n, suppress_stop=True, seed=1)
    torch.cuda.synchronize()
    print(f"beam-sample 3 beams: {n} steps, {(time.time()-t0)/n*1e6:.0f} us/step", flush=True)
