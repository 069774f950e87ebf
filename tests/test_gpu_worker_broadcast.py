"""The opt-in load path of a multi-worker server (IXTTS_BROADCAST_LOAD=1): worker 1 reads model_dir, the others receive every
weight over torch.distributed -- the glue tensors as one packed message, the GPT / BigVGAN weights as their packed device arenas
-- and must synthesise exactly what a worker that read the files synthesises.  Two worker processes on the box's one GPU; the
backend is gloo here (RCCL does not form a group of two ranks on one device; the single-rank RCCL call on the raw arenas is
tests/test_gpu_broadcast_path.py), the code path -- `server.default_model_factory` -> `IndexTTS2(weight_broadcast=...)` -- is the
product's."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import synthetic_model_dir as SM
from voice_tts_amd.front import TextNormalizer, TextTokenizer
from voice_tts_amd import server

root = sys.argv[1]
class Same:
    def normalize(self, s):
        return s
tok = TextTokenizer(root + "/bpe.model", TextNormalizer(Same(), Same()))
if os.environ.get("WORKER_ID") != "1" and os.environ.get("IXTTS_BROADCAST_LOAD") == "1":
    # a receiving worker must not need the checkpoints: hide them
    import torch
    real_load = torch.load
    def guarded(path, *a, **k):
        assert not str(path).endswith(("gpt.pth", "s2mel.pth", "bigvgan_generator.pt", "campplus_cn_common.bin", "feat1.pt", "feat2.pt")), path
        return real_load(path, *a, **k)
    torch.load = guarded
m = server.default_model_factory(cfg_path=root + "/config.yaml", model_dir=root, tokenizer=tok, max_seq=256, max_frames=256)
assert m.ready(), m.missing_glue
import torch as _t
_t.manual_seed(0)  # the CFM noise is drawn on the device from the default generator
sr, pcm = m.infer(SM.synthetic_wav_bytes(1.5, 24000), "Hello world, this is a test.", None, num_beams=1, top_k=1, max_mel_tokens=16)
np.save(sys.argv[2], pcm)
print("worker", os.environ.get("WORKER_ID"), "done", pcm.shape, flush=True)
"""


def test_receiving_worker_synthesises_what_the_reading_worker_does(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synthetic_model_dir as SM

    root = str(tmp_path / "model_dir")
    SM.write_model_dir(root)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, IXTTS_BROADCAST_LOAD="1", IXTTS_WORKERS="2", IXTTS_BROADCAST_PORT=str(port), IXTTS_BROADCAST_BACKEND="gloo", IXTTS_WARMUP="0")
    procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path / f"pcm{w}.npy")], env=dict(env, WORKER_ID=str(w)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for w in (1, 2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], "\n".join(o[-2000:] for o in outs)
    a, b = np.load(tmp_path / "pcm1.npy"), np.load(tmp_path / "pcm2.npy")
    assert a.shape == b.shape and a.shape[0] == int(16 * 1.72) * 256
    # same weights, same kernels, greedy decode; the CFM noise is drawn on the device from the default generator (seeded alike in
    # two fresh processes): identical PCM
    assert np.array_equal(a, b)
