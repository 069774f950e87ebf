"""Spike for row N3: how much does the latency-bound decode graph slow down while the compute-bound s2mel stage of another
request runs on a second stream (two host threads)?  Prints decode us/step alone / overlapped and s2mel ms alone / overlapped."""
import sys, threading, time
import torch
sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
import voice_tts_amd.s2mel as S2
from voice_tts_amd.gpt_engine import GptEngine

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=2048, max_batch=B, device=dev).load_state_dict(WR.make_gpt_weights(WR.GPT_CFG, seed=1234))
m = S2.S2Mel(S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234), S2.S2MEL_CFG, device=dev)
g = torch.Generator().manual_seed(1)
n = 1100
lat = torch.randn(1, n, 1280, generator=g).to(dev); codes = torch.randint(0, 8192, (1, n), generator=g).to(dev)
pc = torch.randn(1, 430, 512, generator=g).to(dev); rm = torch.randn(1, 80, 430, generator=g).to(dev); st = torch.randn(1, 192, generator=g).to(dev)
emb = torch.randn(136, 1280, generator=g) * 0.5


def decode_run(stream, out):
    with torch.cuda.stream(stream):
        for b in range(B):
            eng.prefill(b, emb, 0)
        stream.synchronize()
        t0 = time.perf_counter()
        eng.decode(B, 1096, suppress_stop=True)
        stream.synchronize()
        out["decode_us"] = (time.perf_counter() - t0) / 1096 * 1e6


def s2mel_run(stream, out, reps):
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        for _ in range(reps):
            m(lat, codes, torch.tensor([n], device=dev), pc, rm, st, n_timesteps=25)
        stream.synchronize()
        out["s2mel_ms"] = (time.perf_counter() - t0) / reps * 1e3


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
o = {}
decode_run(sa, o); s2mel_run(sb, o, 1)  # warm-up (graphs, MIOpen find)
decode_run(sa, o); s2mel_run(sb, o, 2)
print(f"alone     : decode {o['decode_us']:.0f} us/step (B={B}), s2mel {o['s2mel_ms']:.0f} ms/segment", flush=True)
o2 = {}
ta = threading.Thread(target=decode_run, args=(sa, o2)); tb = threading.Thread(target=s2mel_run, args=(sb, o2, 3))
t0 = time.perf_counter(); tb.start(); ta.start(); ta.join(); tb.join(); wall = time.perf_counter() - t0
print(f"overlapped: decode {o2['decode_us']:.0f} us/step, s2mel {o2['s2mel_ms']:.0f} ms/segment; wall {wall*1e3:.0f} ms for 1096 steps + 3 segments "
      f"(serial would be {(o['decode_us']*1096/1e3 + 3*o['s2mel_ms']):.0f} ms)", flush=True)
