"""Developer timeline of one decode graph (8 steps): where does each kernel's time go?

    IXTTS_TRACE=1 python -m voice_tts_amd.build          # builds voice-tts_amd/libixtts_hip_trace.so
    IXTTS_LIB=voice-tts_amd/libixtts_hip_trace.so python tools/trace_decode.py [ctx]

Every workgroup's wave 0 records wall-clock (100 MHz) at entry / after its dot products / at exit.  Per kernel
instance: ramp = last WG entry - first WG entry; span = last exit - first entry; gap = next instance's first entry
- this instance's last exit (the kernel boundary as the hardware sees it).
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
from voice_tts_amd import _lib
from voice_tts_amd.gpt_engine import GptEngine

assert "trace" in os.environ.get("IXTTS_LIB", ""), "set IXTTS_LIB to the trace build"
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 700
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
L = _lib.lib()
L.ixtts_trace_begin.restype = C.c_int
L.ixtts_trace_begin.argtypes = []
L.ixtts_trace_read.restype = C.c_int
L.ixtts_trace_read.argtypes = [C.c_void_p]
SLOTS, WGS = 2048, 512

W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=2048, max_batch=B, device=dev).load_state_dict(W)
emb = torch.randn(136, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
for b in range(B):
    eng.prefill(b, emb, 0)
n_warm = max(8, (ctx - 137) // 8 * 8)
eng.decode(B, n_warm, suppress_stop=True)
torch.cuda.synchronize()
eng.decode(B, 8, suppress_stop=True)  # make sure the 8-step graph exists before the table is cleared
torch.cuda.synchronize()
_lib.check(L.ixtts_trace_begin(), "trace_begin")
eng.decode(B, 8, suppress_stop=True)
torch.cuda.synchronize()
buf = np.zeros((SLOTS * WGS, 8), dtype=np.uint64)
_lib.check(L.ixtts_trace_read(buf.ctypes.data), "trace_read")
r = buf[buf[:, 0] != 0]
n = C.c_uint(len(r))
kid = (r[:, 0] >> np.uint64(32)).astype(np.int64) - 1
seq = (r[:, 0] & np.uint64(0xffffffff)).astype(np.int64)
t = r[:, 1:5].astype(np.int64)  # entry, inputs arrived, stream reduced, exit
t -= t[:, 0].min()
order = np.argsort(seq, kind="stable")
kid, t, seq = kid[order], t[order], seq[order]
names = {0: "qkv", 1: "out-proj", 2: "fc", 3: "head", 4: "mlp-proj", 5: "attn"}
# split into kernel instances by launch id
bounds = [0] + [i for i in range(1, len(kid)) if seq[i] != seq[i - 1]] + [len(kid)]
inst = []
for a, b in zip(bounds[:-1], bounds[1:]):
    tt = t[a:b]
    inst.append(dict(kid=int(kid[a]), n=b - a, first=tt[:, 0].min(), last_entry=tt[:, 0].max(), end=tt[:, 3].max(),
                     wg_in=(tt[:, 1] - tt[:, 0]).mean(), wg_mid=(tt[:, 2] - tt[:, 0]).mean(), wg_tot=(tt[:, 3] - tt[:, 0]).mean(),
                     wg_max=(tt[:, 3] - tt[:, 0]).max()))
print(f"records {n.value}, kernel instances {len(inst)}, context ~{137 + n_warm}, B={B}; unit = us (10 ns ticks)")
print(f"{'kernel':10s} {'inst':>5s} {'WGs':>5s} {'ramp':>7s} {'wg->x':>7s} {'wg->dot':>8s} {'wg tot':>7s} {'wg max':>7s} {'span':>7s} {'gap>next':>9s} {'period':>7s}")
for k in sorted(names):
    sel = [i for i, x in enumerate(inst) if x["kid"] == k and i + 1 < len(inst)]
    if not sel:
        continue
    f = lambda key: np.mean([inst[i][key] for i in sel]) / 100.0
    ramp = np.mean([inst[i]["last_entry"] - inst[i]["first"] for i in sel]) / 100.0
    span = np.mean([inst[i]["end"] - inst[i]["first"] for i in sel]) / 100.0
    gap = np.mean([inst[i + 1]["first"] - inst[i]["end"] for i in sel]) / 100.0
    period = np.mean([inst[i + 1]["first"] - inst[i]["first"] for i in sel]) / 100.0
    print(f"{names[k]:10s} {len(sel):5d} {int(np.mean([inst[i]['n'] for i in sel])):5d} {ramp:7.2f} {f('wg_in'):7.2f} {f('wg_mid'):8.2f} {f('wg_tot'):7.2f} {f('wg_max'):7.2f} "
          f"{span:7.2f} {gap:9.2f} {period:7.2f}")
tot = (inst[-1]["end"] - inst[0]["first"]) / 100.0
print(f"traced span of 8 steps: {tot:.1f} us -> {tot / 8:.1f} us/step")
