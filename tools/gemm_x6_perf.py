"""csrc/gemm_x6.hip against torch's fp32 library GEMM (TunableOp file on, as the s2mel glue runs it) on the DiT / WaveNet shapes
at one segment (M = 2 x 2322 rows) and two batched segments: us per call, fp32-equivalent TFLOP/s, per tile shape."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voice_tts_amd import gemm as G  # noqa: E402
from voice_tts_amd.s2mel import use_tuned_gemms  # noqa: E402

dev = torch.device("cuda:0")
use_tuned_gemms()


def bench(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


shapes = [("wqkv", 512, 1536), ("wo", 512, 512), ("w1|w3", 512, 3072), ("w2", 1536, 512), ("wavenet tap", 512, 1024), ("res_skip", 512, 1024),
          ("t_embed", 256, 512), ("adaln proj", 512, 27648)]
for M in [int(a) for a in (sys.argv[1:] or ["4644", "9288"])]:
    for name, K, N in shapes:
        if name == "adaln proj":
            Mx = 2  # (one row per batch entry)
        else:
            Mx = M
        x = torch.randn(Mx, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        pl = G.PackedLinear(w, b)
        out = torch.empty(Mx, N, device=dev)
        fl = 2.0 * Mx * N * K
        t_lib = bench(lambda: torch.nn.functional.linear(x, w, b))
        line = f"M={Mx:5d} {name:12s} K={K:4d} N={N:5d}: library {t_lib:7.1f} us ({fl / t_lib / 1e6:6.1f} TF)"
        for tile in (2, 3, 4, 5):
            try:
                planes = G.split(x)
                t = bench(lambda: G.linear(planes, pl, out=out, tile=tile))
                t2 = bench(lambda: G.linear(x, pl, out=out, tile=tile))
                line += f" | tile{tile} {t:6.1f} us ({fl / t / 1e6:5.1f} TF), with split {t2:6.1f}"
            except Exception as e:
                line += f" | tile{tile} n/a"
        print(line, flush=True)
