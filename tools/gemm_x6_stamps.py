"""Phase stamps of one workgroup of csrc/gemm_x6.hip (developer build path tile 32 / 33): cycles per phase and step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voice_tts_amd import gemm as G, _lib
dev = torch.device("cuda:0")
M, K, N = 4644, 512, 3072
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
pl = G.PackedLinear(w); planes = G.split(x); out = torch.empty(M, N, device=dev)
for tile in (32, 33):
    dbg = torch.zeros(16 * 8 + 4 * 1024, dtype=torch.int64, device=dev)
    pl.bias = dbg  # (the stamps buffer rides in the bias slot)
    for _ in range(3):
        G.linear(planes, pl, out=out, tile=tile)
    torch.cuda.synchronize()
    full = dbg.cpu().numpy()
    d = full[:128].reshape(16, 8)
    nwg = (19 if tile == 32 else 37) * 24
    w = full[128:128 + 4 * nwg].reshape(nwg, 4).astype('float64') * 0.01  # us
    t0 = w[:, 0].min()
    import numpy as np
    print(f'  workgroups {nwg}: entry spread {w[:,0].max()-t0:.1f} us; prologue {np.mean(w[:,1]-w[:,0]):.1f} (max {np.max(w[:,1]-w[:,0]):.1f}); main loop mean {np.mean(w[:,2]-w[:,1]):.1f} min {np.min(w[:,2]-w[:,1]):.1f} max {np.max(w[:,2]-w[:,1]):.1f}; epilogue mean {np.mean(w[:,3]-w[:,2]):.1f} max {np.max(w[:,3]-w[:,2]):.1f}; last exit at {w[:,3].max()-t0:.1f} us; wg0 loop {w[0,2]-w[0,1]:.1f}')
    names = ["issue", "frag reads", "mfma", "vmcnt wait", "barrier"]
    print(f"tile {tile}: cycles per phase (steps 2..13 of workgroup 0, wave 0); clock = d(memtime)/d(memrealtime)*100 MHz")
    for s_ in range(2, 14):
        row = d[s_]
        print(f"  step {s_:2d}: " + "  ".join(f"{n} {int(row[k + 1] - row[k]):5d}" for k, n in enumerate(names)) + f"   | step total {int(d[s_ + 1][0] - row[0]):5d}")
    clk = (d[13][0] - d[2][0]) / max(1, (d[13][7] - d[2][7])) * 100
    print(f"  shader clock over these steps: {clk:.0f} MHz")
