"""Conditioning encoders on a 249-frame prompt (production sizes): ms per request."""
import sys, time, torch
sys.path.insert(0, ".")
import voice_tts_amd.conditioning as CD
import voice_tts_amd.s2mel as S2
dev = torch.device("cuda:0")
S2.use_tuned_gemms()
m = CD.Conditioning(CD.make_cond_weights(CD.COND_CFG, seed=1234), CD.COND_CFG, device=dev)
x = torch.randn(1, 249, 1024, generator=torch.Generator().manual_seed(1)).to(dev)
ls = torch.tensor([1024], device=dev)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    with torch.no_grad():
        ev = m.merge_emovec(x, x, ls, ls, alpha=1.0)
        c = m.get_conditioning(x.transpose(1, 2), ls)
    torch.cuda.synchronize()
    print(f"conditioning: {(time.time()-t0)*1e3:.2f} ms", flush=True)
