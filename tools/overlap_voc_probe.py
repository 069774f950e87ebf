"""Does the vocoder of segment k overlap usefully with the s2mel stage of segment k+1 (two streams, two host threads)?
Prints each alone and both together."""
import sys, threading, time
import torch
sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
import voice_tts_amd.s2mel as S2
from voice_tts_amd.bigvgan import BigVGAN

dev = torch.device("cuda:0")
m = S2.S2Mel(S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234), S2.S2MEL_CFG, device=dev)
bv = BigVGAN(WR.BIGVGAN_CFG, max_frames=2048, device=dev).load_state_dict(WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234))
g = torch.Generator().manual_seed(1)
n = 1100
lat = torch.randn(1, n, 1280, generator=g).to(dev); codes = torch.randint(0, 8192, (1, n), generator=g).to(dev)
pc = torch.randn(1, 430, 512, generator=g).to(dev); rm = torch.randn(1, 80, 430, generator=g).to(dev); st = torch.randn(1, 192, generator=g).to(dev)
mel = (torch.randn(1, 80, 1892, generator=g) * 2 - 4).clamp(-11.5, 2).to(dev)


def s2mel_run(stream, out, reps):
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        for _ in range(reps):
            m(lat, codes, torch.tensor([n], device=dev), pc, rm, st, n_timesteps=25)
        stream.synchronize()
        out["s2mel_ms"] = (time.perf_counter() - t0) / reps * 1e3


def voc_run(stream, out, reps):
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        for _ in range(reps):
            bv(mel)
        stream.synchronize()
        out["voc_ms"] = (time.perf_counter() - t0) / reps * 1e3


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
o = {}
s2mel_run(sa, o, 1); voc_run(sb, o, 2)
s2mel_run(sa, o, 2); voc_run(sb, o, 3)
print(f"alone   : s2mel {o['s2mel_ms']:.1f} ms, vocoder {o['voc_ms']:.1f} ms", flush=True)
for trial in range(2):
    o2 = {}
    ta = threading.Thread(target=s2mel_run, args=(sa, o2, 1)); tb = threading.Thread(target=voc_run, args=(sb, o2, 1))
    torch.cuda.synchronize(); t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); wall = (time.perf_counter() - t0) * 1e3
    print(f"together: s2mel {o2['s2mel_ms']:.1f} ms, vocoder {o2['voc_ms']:.1f} ms, wall {wall:.1f} ms (serial {o['s2mel_ms'] + o['voc_ms']:.1f})", flush=True)
