"""Kernel mix of the batched-rows path: prefill of a 137-row prompt and the latent pass over 1100 codes (for rocprofv3 --stats)."""
import sys, time, torch
sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
from voice_tts_amd.pipeline import HotPath
dev = torch.device("cuda:0")
hp = HotPath(dtype="bf16", device=dev, max_batch=2, max_seq=1400, max_frames=64)
hp.load(WR.make_gpt_weights(WR.GPT_CFG, seed=1234), WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234))
g = torch.Generator().manual_seed(1)
conds = (torch.randn(34, 1280, generator=g) * 0.5).to(dev)
text = torch.randint(2, 12000, (100,), generator=g)
codes = torch.randint(0, 8192, (1100,), generator=g).numpy()
emb, pad, P = hp.prepare_gpt_inputs(conds, text)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    hp.gpt.prefill(0, emb, pad)
    torch.cuda.synchronize(); t1 = time.time()
    hp.latent(conds, text, codes)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"prefill {1e3*(t1-t0):.2f} ms  latent {1e3*(t2-t1):.2f} ms", flush=True)
