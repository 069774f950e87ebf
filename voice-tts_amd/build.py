"""Ahead-of-time build of libixtts_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m voice_tts_amd.build        # or: __graft_entry__.build()

The library is built IN-TREE (voice-tts_amd/libixtts_hip.so) so that it travels with the
repo snapshot to the GPU box; it is git-ignored.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libixtts_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
TRACE = os.environ.get("IXTTS_TRACE") == "1"  # developer timeline build (tools/trace_decode.py): separate library
if TRACE:
    LIB = os.path.join(HERE, "libixtts_hip_trace.so")
    OBJ = os.path.join(HERE, "build_trace")
VARIANT = os.environ.get("IXTTS_VARIANT")  # developer A/B builds: IXTTS_VARIANT=name IXTTS_EXP="-DX=1" -> libixtts_hip_<name>.so (load it with IXTTS_LIB)
if VARIANT:
    LIB = os.path.join(HERE, f"libixtts_hip_{VARIANT}.so")
    OBJ = os.path.join(HERE, f"build_{VARIANT}")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-Wno-unused-value", "-fno-gpu-rdc",
         # first 16 kernarg dwords arrive in SGPRs with the wave (no s_load round trip before the first global loads)
         "-mllvm", "-amdgpu-kernarg-preload-count=16"] + (["-DIXTTS_TRACE"] if TRACE else []) + os.environ.get("IXTTS_EXP", "").split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path):
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/ixtts_hip.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and (f.endswith((".h", ".hip")) or f.endswith("ixtts_hip.h")):
            if f.endswith(".h") or os.path.samefile(p, path):
                h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    procs = []
    for src in _sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        stamp = op + ".sha"
        dg = _digest(sp)
        objs.append(op)
        if not force and os.path.exists(op) and os.path.exists(stamp) and open(stamp).read() == dg:
            continue
        cmd = [HIPCC, *FLAGS, "-c", sp, "-o", op]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), stamp, dg, src))
    failed = []
    for p, stamp, dg, src in procs:
        if p.wait() != 0:
            failed.append(src)
        else:
            open(stamp, "w").write(dg)
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    if procs or not os.path.exists(LIB) or force:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
