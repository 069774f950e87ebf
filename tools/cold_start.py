#!/usr/bin/env python
"""Where does the FIRST request of a fresh worker spend its time?  (182 s on the driver's fresh box.)

Runs every stage of the bench request twice, in order, and prints the wall time of the first and of the second call.
Run it with a fresh HOME (MIOpen's per-user kernel cache and find-db live under it) to emulate a box that never ran the repo:

    HOME=$(mktemp -d) MIOPEN_ENABLE_LOGGING_CMD=1 python tools/cold_start.py 2> gpurun_out/r03/cold_start.err
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T0 = time.perf_counter()

import torch  # noqa: E402


def say(msg):
    print(f"[cold {time.perf_counter() - T0:8.2f}s] {msg}", flush=True)


say("torch imported")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)
torch.cuda.synchronize()
say("device initialised")

import voice_tts_amd.conditioning as CD  # noqa: E402
import voice_tts_amd.s2mel as S2  # noqa: E402
import voice_tts_amd.weights as WR  # noqa: E402
from voice_tts_amd.pipeline import HotPath  # noqa: E402


def mapped():
    """file -> mapped bytes of this process (shared objects, code-object files, databases the libraries mmap)"""
    m = {}
    for ln in open("/proc/self/maps"):
        f = ln.split()
        if len(f) >= 6 and f[5].startswith("/"):
            a, b = f[0].split("-")
            m[f[5]] = m.get(f[5], 0) + int(b, 16) - int(a, 16)
    return m


def io_counters():
    return {k.strip(): int(v) for k, v in (ln.split(":") for ln in open("/proc/self/io"))}


def opened():
    s = set()
    for fd in os.listdir("/proc/self/fd"):
        try:
            s.add(os.readlink(f"/proc/self/fd/{fd}"))
        except OSError:
            pass
    return s


def timed(name, fn, n=2):
    out = None
    for i in range(n):
        torch.cuda.synchronize()
        m0, io0, fd0 = mapped(), io_counters(), opened()
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        m1, io1, fd1 = mapped(), io_counters(), opened()
        say(f"{name}: call {i} {1e3 * dt:.1f} ms; read() {1e-6 * (io1['rchar'] - io0['rchar']):.1f} MB, from storage {1e-6 * (io1['read_bytes'] - io0['read_bytes']):.1f} MB")
        for f in sorted(set(m1) - set(m0)):
            say(f"    newly mapped: {f} ({1e-6 * m1[f]:.1f} MB mapped, file {1e-6 * os.path.getsize(f) if os.path.exists(f) else 0:.1f} MB)")
        for f in sorted(fd1 - fd0):
            if f.startswith("/") and not f.startswith("/dev") and not f.startswith("/proc"):
                say(f"    left open: {f}")
    return out


n_codes, n_tok = int(os.environ.get("CODES", "1100")), 100
P = 34 + n_tok + 2 + 1
frames = int(n_codes * 1.72)
hp = HotPath(dtype="bf16", device=dev, max_batch=2, max_seq=P + n_codes + 64, max_frames=frames)
Wg = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
Wb = WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234)
say("weights generated")
hp.load(Wg, Wb)
torch.cuda.synchronize()
say("gpt + bigvgan loaded")
hp.attach_s2mel(S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234))
hp.attach_conditioning(CD.make_cond_weights(CD.COND_CFG, seed=1234))
torch.cuda.synchronize()
say("glue attached")

g = torch.Generator().manual_seed(100)
spk = torch.randn(1, 249, 1024, generator=g).to(dev)
pc = torch.randn(1, 430, 512, generator=g).to(dev)
ref_mel = (torch.randn(1, 80, 430, generator=g) * 2 - 4).clamp(-11.5, 2).to(dev)
style = torch.randn(1, 192, generator=g).to(dev)
text = torch.randint(2, 12000, (n_tok,), generator=g)

cl = timed("conditioning (conformer + perceiver, torch)", lambda: hp.conds_from_prompt(spk, None, 1.0))
cl = cl * (0.5 / cl.std().clamp_min(1e-6))
emb, pad, _ = hp.prepare_gpt_inputs(cl, text)
timed("prefill x2", lambda: [hp.gpt.prefill(b, emb, pad) for b in range(2)])
timed("decode 64 steps B=2 (graph capture on call 0)", lambda: hp.gpt.decode(2, 64, repetition_penalty=10.0, suppress_stop=True))
codes = torch.randint(0, 8192, (n_codes,), generator=g).numpy()
lat = timed("latent pass", lambda: hp.latent(cl, text, codes))
m = hp.s2mel_model
codes_t = torch.as_tensor(codes, dtype=torch.long, device=dev).reshape(1, -1)
timed("s2mel.gpt_layer", lambda: m.gpt_layer(lat.reshape(1, n_codes, -1)))
timed("s2mel.vq2emb (conv1d k=1)", lambda: m.vq2emb(codes_t))
x = m.vq2emb(codes_t) + m.gpt_layer(lat.reshape(1, n_codes, -1))
timed("s2mel.length_regulator (conv1d k=3 x4, group_norm, mish)", lambda: m.length_regulator(x, torch.tensor([frames], device=dev)))
timed("s2mel 1 Euler step", lambda: hp.s2mel(lat, codes, pc, ref_mel, style, n_timesteps=1))
mel = timed("s2mel 25 Euler steps", lambda: hp.s2mel(lat, codes, pc, ref_mel, style))
timed("bigvgan", lambda: hp.vocode(mel.clamp(-11.5, 2.0)).to(torch.int16).cpu())
say("done")
