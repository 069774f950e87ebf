"""r02 left a note: replaying the conditioning passes from a hipGraph "returned a different emotion vector" for requests without
a separate emotion prompt.  This probe captures `Conditioning._encode_eager` per case in a torch.cuda.CUDAGraph with static
input buffers and compares replay with eager, for (a) no emotion prompt (ec is sc), (b) a separate emotion prompt, and
(c) the suspected cause: ONE graph captured with two distinct input buffers (so it always runs merge_emovec on both) replayed
for a request without an emotion prompt while the emotion buffer still holds the previous request's features."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_tts_amd.conditioning as CD  # noqa: E402

dev = torch.device("cuda:0")
cd = CD.Conditioning(CD.make_cond_weights(CD.COND_CFG, seed=1234), CD.COND_CFG, device=dev)
g = torch.Generator().manual_seed(1)
spk = torch.randn(1, 249, 1024, generator=g).to(dev)
emo = torch.randn(1, 249, 1024, generator=g).to(dev)
other = torch.randn(1, 249, 1024, generator=g).to(dev)
ls = torch.tensor([1024], device=dev)


def eager(sc, ec, alpha):
    with torch.no_grad():
        return cd._encode_eager(sc, ec, alpha, ls, ls)


def capture(two_buffers):
    s_sc, s_ec = spk.clone(), (emo.clone() if two_buffers else None)
    with torch.no_grad():
        for _ in range(3):  # warm-up on a side stream, as torch requires
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                cd._encode_eager(s_sc, s_ec if two_buffers else s_sc, 0.7 if two_buffers else 1.0, ls, ls)
            torch.cuda.current_stream().wait_stream(st)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = cd._encode_eager(s_sc, s_ec if two_buffers else s_sc, 0.7 if two_buffers else 1.0, ls, ls)
    return gr, s_sc, s_ec, out


def maxdiff(a, b):
    return max(float((x - y).abs().max()) for x, y in zip(a, b))


g1, s1, _, o1 = capture(False)
s1.copy_(spk)
g1.replay()
torch.cuda.synchronize()
print("(a) no emotion prompt, graph vs eager: max|diff| cond32/emovec =", maxdiff(o1, eager(spk, spk, 1.0)))
s1.copy_(other)
g1.replay()
torch.cuda.synchronize()
print("(a') same graph, another speaker prompt:", maxdiff(o1, eager(other, other, 1.0)))
g2, s2, e2, o2 = capture(True)
s2.copy_(spk); e2.copy_(emo)
g2.replay()
torch.cuda.synchronize()
print("(b) separate emotion prompt (alpha 0.7), graph vs eager:", maxdiff(o2, eager(spk, emo, 0.7)))
# (c) the two-buffer graph replayed for a request WITHOUT an emotion prompt: the emotion buffer is stale unless the caller refreshes it,
# and alpha is baked into the captured kernels' arguments (0.7), while eager uses merge_emovec(spk, spk, 1.0) = the speaker's own vector
s2.copy_(other)
g2.replay()
torch.cuda.synchronize()
ref = eager(other, other, 1.0)
print("(c) two-buffer graph, request without emotion prompt, emotion buffer NOT refreshed: emovec max|diff| =", float((o2[1] - ref[1]).abs().max()))
e2.copy_(other)
g2.replay()
torch.cuda.synchronize()
print("(c') emotion buffer refreshed with the speaker features:", float((o2[1] - ref[1]).abs().max()), "(alpha 0.7 baked in: base + 0.7 (base - base) = base)")
for name, fn in (("eager", lambda: eager(spk, spk, 1.0)), ("graph", lambda: g1.replay())):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per request")
