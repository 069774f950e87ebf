import sys, time, ctypes, torch
sys.path.insert(0, ".")
import voice_tts_amd.weights as WR
from voice_tts_amd.gpt_engine import GptEngine
from voice_tts_amd import _lib
dev = torch.device("cuda:0")
eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=2048, max_batch=3, device=dev).load_state_dict(WR.make_gpt_weights(WR.GPT_CFG, seed=1234))
emb = torch.randn(136, 1280, generator=torch.Generator().manual_seed(1)) * 0.5
eng.prefill(0, emb, 0)
eng.beam_begin(3)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 64)()
import numpy as np
acc = []
for it in range(40):
    eng.beam_decode(8, suppress_stop=True, seed=1)
    torch.cuda.synchronize()
    L.ixtts_debug_beam_ts(buf)
    acc.append(np.array(list(buf), dtype=np.int64))
a = np.stack(acc[5:])
def d(i, j): return float(np.median(a[:, j] - a[:, i])) * 10 / 1000  # 100 MHz -> us
print("cand: entry->loads %.2f  ->lse %.2f  ->penalty %.2f  ->topk %.2f  ->topp %.2f" % (d(0,1), d(1,2), d(2,3), d(3,4), d(4,5)))
print("cand end -> step entry %.2f" % d(5,10))
print("step: entry->loads %.2f  ->list %.2f  ->draw %.2f  ->sort %.2f  ->process %.2f  ->sync %.2f  ->stores issued %.2f -> end %.2f ; other-wave loads done at %.2f after entry" % (d(10,11), d(11,12), d(12,13), d(13,14), d(14,15), d(15,16), d(16,17), d(17,18), d(10,20)))
print("topk: radix %.2f (pass0 add %.2f pick %.2f | pass1 add %.2f pick %.2f)  collect %.2f  rank %.2f  sync %.2f   pool=%d" % (d(30,31), d(30,35), d(35,36), d(36,37), d(37,38), d(31,32), d(32,33), d(33,34), int(np.median(a[:,40]))))
