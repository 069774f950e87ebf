import sys, time, torch
sys.path.insert(0, ".")
import voice_tts_amd.s2mel as S2
dev = torch.device("cuda:0")
m = S2.S2Mel(S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234), S2.S2MEL_CFG, device=dev)
g = torch.Generator().manual_seed(1)
n = 1100
lat = torch.randn(1, n, 1280, generator=g).to(dev); codes = torch.randint(0, 8192, (1, n), generator=g).to(dev)
pc = torch.randn(1, 430, 512, generator=g).to(dev); rm = torch.randn(1, 80, 430, generator=g).to(dev); st = torch.randn(1, 192, generator=g).to(dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    mel = m(lat, codes, torch.tensor([n], device=dev), pc, rm, st, n_timesteps=steps)
    torch.cuda.synchronize(); print(f"s2mel {steps} steps: {(time.time()-t0)*1e3:.1f} ms", mel.shape, flush=True)
