"""Probe: fp32 GEMM emulated as one bf16 GEMM over a K-concatenated 3-way split (6 products), library kernels only."""
import sys
import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)


def ev(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def split3(x):
    h = x.to(torch.bfloat16)
    r = x - h.float()
    m = r.to(torch.bfloat16)
    l = (r - m.float()).to(torch.bfloat16)
    return h, m, l


M = 4644
for K, N in ((512, 1536), (512, 3072), (1536, 512), (512, 512)):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    ref = (A.double() @ W.double().t())
    t32 = ev(lambda: torch.mm(A, W.t()))
    e32 = ((A @ W.t()).double() - ref).abs().max().item()
    ah, am, al = split3(A)
    wh, wm, wl = split3(W)
    # 6 products: hh, hm, mh, mm, hl, lh
    A6 = torch.cat([ah, ah, am, am, ah, al], 1).contiguous()
    W6 = torch.cat([wh, wm, wh, wm, wl, wh], 1).contiguous()
    A3 = torch.cat([ah, ah, am], 1).contiguous()
    W3 = torch.cat([wh, wm, wh], 1).contiguous()
    try:
        f6 = lambda: torch.mm(A6, W6.t(), out_dtype=torch.float32)
        o6 = f6()
        t6 = ev(f6)
        e6 = (o6.double() - ref).abs().max().item()
        f3 = lambda: torch.mm(A3, W3.t(), out_dtype=torch.float32)
        t3 = ev(f3)
        e3 = (f3().double() - ref).abs().max().item()
    except Exception as e:  # noqa: BLE001
        print("out_dtype mm failed:", repr(e)[:300])
        break
    tb = ev(lambda: torch.mm(ah, wh.t()))
    tsplit = ev(lambda: split3(A))
    print(f"K={K} N={N}: fp32 {t32:.1f} us (err {e32:.2e}) | bf16x6 {t6:.1f} us (err {e6:.2e}) | bf16x3 {t3:.1f} us (err {e3:.2e}) | plain bf16 {tb:.1f} us | split3(A) torch {tsplit:.1f} us", flush=True)
