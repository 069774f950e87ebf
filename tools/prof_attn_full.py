"""Run the DiT attention kernel on the production shape (B=2, H=8, T=2322) a few times (for rocprofv3 --pmc / --stats)."""
import sys, time, torch
sys.path.insert(0, ".")
from voice_tts_amd.s2mel import attn_full
dev = torch.device("cuda:0")
B, H, T = 2, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 2322
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B, T, 3, H, 64, generator=g).to(dev)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
for _ in range(3):
    o = attn_full(q, k, v)
torch.cuda.synchronize()
t0 = time.time()
n = 20
for _ in range(n):
    o = attn_full(q, k, v)
torch.cuda.synchronize()
us = (time.time() - t0) / n * 1e6
print(f"attn_full B={B} H={H} T={T}: {us:.1f} us  {4.0 * B * H * T * T * 64 / us / 1e6:.1f} TFLOP/s", flush=True)
