"""Row N3 host logic: the decode scheduler against a scripted engine (no GPU)."""
import numpy as np
import torch

from voice_tts_amd.scheduler import BeamGroupScheduler, DecodeScheduler, Segment

STOP = 99


class ScriptedEngine:
    """Slot b emits script[b][k] at its k-th step after prefill; the script is chosen by the prompt's first element."""

    def __init__(self, scripts, max_batch, max_seq=64):
        self.scripts, self.max_batch, self.max_seq = scripts, max_batch, max_seq
        self.seq = [None] * max_batch
        self.calls = []

    def prefill(self, b, embeds, pad):
        key = int(embeds[0, 0])
        self.seq[b] = dict(key=key, out=[], rows=embeds.shape[0] + 1)
        self.calls.append(("prefill", b, key))

    def decode(self, n_active, n_steps, **kw):
        assert all(self.seq[b] is not None for b in range(n_active)), "every active slot needs a prompt"
        self.calls.append(("decode", n_active, n_steps))
        for b in range(n_active):
            s = self.seq[b]
            for _ in range(n_steps):
                done = STOP in s["out"] and not kw.get("suppress_stop")
                script = self.scripts.get(s["key"], [])
                s["out"].append(STOP if done or len(s["out"]) >= len(script) else script[len(s["out"])])
                s["rows"] += 1
                assert s["rows"] < self.max_seq, "a slot ran off the end of its cache"

    def read(self, b):
        out = np.asarray(self.seq[b]["out"], np.int32)
        return out, bool((out == STOP).any())


def _seg(key, max_new, rows=5):
    e = torch.zeros(rows, 4)
    e[0, 0] = key
    return Segment(request=key // 10, index=key % 10, embeds=e, n_left_pad=0, max_new=max_new)


def test_continuous_batching_refills_and_trims():
    scripts = {11: [1, 2, STOP], 12: [3] * 20, 21: [4, 5, 6, 7, STOP], 22: [8, STOP], 31: [9] * 7}
    eng = ScriptedEngine(scripts, max_batch=2)
    got = {}
    sched = DecodeScheduler(eng, max_batch=2, stop_token=STOP, sync_every=4)
    stats = sched.run([_seg(k, 10) for k in (11, 12, 21, 22, 31)], lambda seg, ids: got.__setitem__((seg.request, seg.index), ids.tolist()))
    assert got == {(1, 1): [1, 2, STOP], (1, 2): [3] * 10, (2, 1): [4, 5, 6, 7, STOP], (2, 2): [8, STOP], (3, 1): [9] * 7 + [STOP]}
    assert stats["refills"] == 3 and stats["busy_slot_steps"] <= stats["slot_steps"]
    # slot 0 was refilled while slot 1 (the 20-token script, cut at max_new=10) kept going
    assert [c for c in eng.calls if c[0] == "prefill"][:3] == [("prefill", 0, 11), ("prefill", 1, 12), ("prefill", 0, 21)]


def test_fixed_length_and_idle_slot_parking():
    scripts = {1: [5, STOP], 2: [7] * 200, 3: [5, STOP]}
    eng = ScriptedEngine(scripts, max_batch=2, max_seq=64)
    got = {}
    DecodeScheduler(eng, max_batch=2, stop_token=STOP, sync_every=8).run([_seg(1, 4, rows=15), _seg(2, 20)], lambda s, ids: got.__setitem__(s.index, ids.tolist()),
                                                                          fixed_length=True)
    assert got[1] == [5, STOP, STOP, STOP] and got[2] == [7] * 20  # fixed length: nothing is trimmed
    # the longest expected decode goes first: the 20-token segment takes slot 0 although it was submitted second
    assert [c for c in eng.calls if c[0] == "prefill"][:2] == [("prefill", 0, 2), ("prefill", 1, 1)]
    # one long sequence in slot 1 while slot 0 idles after a sequence that stopped early: the idle slot is re-parked before it overruns max_seq
    eng = ScriptedEngine(scripts, max_batch=2, max_seq=30)
    got = {}
    DecodeScheduler(eng, max_batch=2, stop_token=STOP, sync_every=8).run([_seg(3, 22, rows=15), _seg(2, 20)], lambda s, ids: got.__setitem__(s.index, ids.tolist()))
    assert got[3] == [5, STOP] and got[2] == [7] * 20
    assert any(c[0] == "prefill" and c[2] == 0 for c in eng.calls), "the idle slot was parked on a stub prompt"


def test_longest_first_is_stable_and_keeps_beam_streams():
    from voice_tts_amd.scheduler import longest_first

    segs = [_seg(11, 10, rows=5), _seg(12, 30, rows=5), _seg(13, 10, rows=9), _seg(14, 10, rows=5)]
    assert [s.index for s in longest_first(segs)] == [2, 3, 1, 4]  # cap first, then prompt rows; ties keep the submission order


class ScriptedBeamEngine:
    """Beam groups against a script: group g holds the segment whose key is its prompt's first element; the segment emits
    script[key][k] at its k-th step and its scorer is `done` when the script says STOP."""

    def __init__(self, scripts, max_batch, max_seq=64):
        self.scripts, self.max_batch, self.max_seq = scripts, max_batch, max_seq
        self.slot_key, self.groups, self.calls = {}, {}, []

    def prefill(self, b, embeds, pad):
        self.slot_key[b] = (int(embeds[0, 0]), embeds.shape[0] + 1)

    def beam_begin(self, num_beams, group=0, rng_stream=0):
        key, rows = self.slot_key[group * num_beams]
        self.groups[group] = dict(key=key, out=[], rows=rows, parked=False, stream=rng_stream)
        self.calls.append(("begin", group, key, rng_stream))

    def beam_park(self, group):
        self.groups[group]["parked"] = True
        self.calls.append(("park", group))

    def beam_decode(self, n_steps, groups=1, suppress_stop=False, **kw):
        self.calls.append(("decode", groups, n_steps))
        for g in range(groups):
            s = self.groups[g]  # every group under the highest live one has been begun at least once
            if s["parked"]:
                continue
            for _ in range(n_steps):
                if STOP in s["out"] and not suppress_stop:
                    break  # scorer done: later steps are no-ops
                script = self.scripts[s["key"]]
                s["out"].append(script[len(s["out"])] if len(s["out"]) < len(script) else 7)
                s["rows"] += 1
                assert s["rows"] < self.max_seq

    def beam_read(self, max_new, group=0):
        out = list(self.groups[group]["out"])
        done = STOP in out
        ids = out[: out.index(STOP)] if done else out
        if len(ids) < max_new:
            ids = ids + [STOP]  # finalize: + eos when there is room
        return np.asarray(ids[:max_new + 1], np.int32), done, -1.0


def test_beam_groups_refill_park_and_streams():
    scripts = {11: [1, 2, STOP], 12: [3] * 40, 21: [4, 5, 6, 7, STOP], 22: [8, STOP], 31: [9] * 30}
    eng = ScriptedBeamEngine(scripts, max_batch=7)  # 7 slots, 3 beams: two groups
    got = {}
    sched = BeamGroupScheduler(eng, 3, sync_every=4)
    assert sched.max_groups == 2
    stats = sched.run([_seg(k, 10) for k in (11, 12, 21, 22, 31)], lambda seg, ids, score: got.__setitem__((seg.request, seg.index), ids.tolist()))
    assert got == {(1, 1): [1, 2, STOP], (1, 2): [3] * 10, (2, 1): [4, 5, 6, 7, STOP], (2, 2): [8, STOP], (3, 1): [9] * 10}
    begins = [c for c in eng.calls if c[0] == "begin"]
    # random stream = position in the submission order, whatever group the segment lands in
    assert [(c[2], c[3]) for c in begins] == [(11, 0), (12, 1), (21, 2), (22, 3), (31, 4)]
    assert begins[2][1] == 0 and stats["refills"] == 3  # group 0 was refilled while group 1 kept going
    # when the queue is empty, a finished group under a live one is parked (a no-op in the beam kernels), not stepped on
    eng2 = ScriptedBeamEngine({1: [5, STOP], 2: [6] * 40}, max_batch=6)
    got2 = {}
    BeamGroupScheduler(eng2, 3, sync_every=4).run([_seg(1, 12), _seg(2, 12)], lambda seg, ids, score: got2.__setitem__(seg.index, ids.tolist()))
    assert got2 == {1: [5, STOP], 2: [6] * 12}
    assert ("park", 0) in eng2.calls and eng2.calls.index(("park", 0)) > eng2.calls.index(("decode", 2, 4))
    # a register engine (<= 4 slots) holds one group: the segments in turn, as the reference runs them
    eng1 = ScriptedBeamEngine(scripts, max_batch=4)
    got1 = {}
    BeamGroupScheduler(eng1, 3, sync_every=4).run([_seg(k, 10) for k in (11, 22)], lambda seg, ids, score: got1.__setitem__(seg.index, ids.tolist()))
    assert got1 == {1: [1, 2, STOP], 2: [8, STOP]} and all(c[1] == 1 for c in eng1.calls if c[0] == "decode")
