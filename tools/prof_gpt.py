import sys, torch
sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
from voice_tts_amd.gpt_engine import GptEngine
dev = torch.device("cuda:0")
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
P = int(sys.argv[3]) if len(sys.argv) > 3 else 137
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
eng = GptEngine(WR.GPT_CFG, dtype=dtype, max_seq=2048, max_batch=B, device=dev).load_state_dict(W)
emb = torch.randn(P - 1, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
for b in range(B):
    eng.prefill(b, emb, 0)
eng.decode(B, steps, suppress_stop=True)
torch.cuda.synchronize()
