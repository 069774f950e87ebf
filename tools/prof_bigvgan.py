"""BigVGAN forward on 1892 frames a few times (for rocprofv3 --stats / --pmc)."""
import sys, time, torch
sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
from voice_tts_amd.bigvgan import BigVGAN
dev = torch.device("cuda:0")
import os
m = BigVGAN(WR.BIGVGAN_CFG, max_frames=2048, fast_sin=os.environ.get("IXTTS_FAST_SIN") == "1", device=dev).load_state_dict(WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1892
mel = (torch.randn(1, 80, F, generator=torch.Generator().manual_seed(6)) * 2 - 4).clamp(-11.5, 2).to(dev)
for _ in range(2):
    w = m(mel)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3):
    w = m(mel)
torch.cuda.synchronize()
ms = (time.time() - t0) / 3 * 1e3
print(f"bigvgan F={F}: {ms:.2f} ms  {m.flops(1, F)/ms/1e9:.1f} TFLOP/s", flush=True)
