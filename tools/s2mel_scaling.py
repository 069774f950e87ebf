"""s2mel (length regulator + 25-step CFM / DiT) time against segment length: is the stage GPU-bound at the short segments of the
mixed workload?  Wall time per call (synchronised) and the time the HOST needs to issue the same call (no sync until the end of 4
back-to-back calls: if 4 calls take 4 x one call, the GPU is the bound; if the issue time equals the wall time, the host is)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_tts_amd.s2mel as S2  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
W = {k: v.to(dev) for k, v in S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234).items()}
m = S2.S2Mel(W, S2.S2MEL_CFG, device=dev)
g = torch.Generator().manual_seed(5)
Tref = 430
pc = torch.randn(1, Tref, 512, generator=g).to(dev)
ref_mel = (torch.randn(1, 80, Tref, generator=g) * 2 - 5).to(dev)
style = torch.randn(1, 192, generator=g).to(dev)
for n in [int(a) for a in sys.argv[1:]] or (1100, 800, 550, 300, 150):
    lat = torch.randn(1, n, 1280, generator=g).to(dev) * 0.3
    codes = torch.randint(0, 8192, (1, n), generator=g).to(dev)
    lens = torch.tensor([n], device=dev)
    noise = torch.randn(1, 80, Tref + int(n * 1.72), generator=g)
    # (a) the default heuristic at this length, (b) the recorded winners re-keyed to it (s2mel.extend_tuned_gemms): same mel?
    os.environ["IXTTS_TUNED_ANY_LENGTH"] = "0"
    for _ in range(2):
        mel0 = m(lat, codes, lens, pc, ref_mel, style, noise=noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m(lat, codes, lens, pc, ref_mel, style, noise=noise)
    torch.cuda.synchronize()
    t_def = time.perf_counter() - t0
    os.environ["IXTTS_TUNED_ANY_LENGTH"] = "1"
    for _ in range(2):
        mel1 = m(lat, codes, lens, pc, ref_mel, style, noise=noise)
    torch.cuda.synchronize()
    print(f"codes {n:5d}: default heuristic {t_def * 1e3:7.1f} ms; re-keyed winners: mel max|diff| {float((mel1 - mel0).abs().max()):.2e} of {float(mel0.abs().max()):.2f}", flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m(lat, codes, lens, pc, ref_mel, style)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_one = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(4):
        m(lat, codes, lens, pc, ref_mel, style)
    torch.cuda.synchronize()
    t_four = (time.perf_counter() - t0) / 4
    T = Tref + int(n * 1.72)
    print(f"codes {n:5d}  T {T:5d}: one call {t_one * 1e3:7.1f} ms (host returned after {t_issue * 1e3:7.1f}), back to back {t_four * 1e3:7.1f} ms  "
          f"-> {t_four * 1e3 / T * 1000:6.1f} us per frame", flush=True)
