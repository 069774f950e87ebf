"""Two independent B=1 decode chains on two streams against one B=2 chain: does interleaving two latency-bound chains beat batching?"""
import sys, time, threading, torch
sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
from voice_tts_amd.gpt_engine import GptEngine
dev = torch.device("cuda:0")
W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
emb = torch.randn(136, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
N = 800
e2 = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=2048, max_batch=2, device=dev).load_state_dict(W)
for b in range(2): e2.prefill(b, emb, 0)
e2.decode(2, 64, suppress_stop=True); torch.cuda.synchronize(); t0 = time.time()
e2.decode(2, N, suppress_stop=True); torch.cuda.synchronize()
print(f"one chain, B=2: {(time.time()-t0)/N*1e6:.1f} us per step (2 sequences)", flush=True)
engs = [GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=2048, max_batch=1, device=dev).load_state_dict(W) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(i, n):
    with torch.cuda.stream(streams[i]):
        engs[i].decode(1, n, suppress_stop=True)
        streams[i].synchronize()
for i in range(2):
    with torch.cuda.stream(streams[i]):
        engs[i].prefill(0, emb, 0)
        engs[i].decode(1, 64, suppress_stop=True)
torch.cuda.synchronize()
t0 = time.time(); run(0, N); one = (time.time() - t0) / N * 1e6
print(f"one chain, B=1: {one:.1f} us per step", flush=True)
ths = [threading.Thread(target=run, args=(i, N)) for i in range(2)]
torch.cuda.synchronize(); t0 = time.time()
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
print(f"two chains, B=1 each, two streams: {(time.time()-t0)/N*1e6:.1f} us per step pair (2 sequences)", flush=True)
