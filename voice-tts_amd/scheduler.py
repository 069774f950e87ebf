"""Cross-segment / cross-request decode batching.

The reference serialises everything behind one lock and decodes one segment at a time (`server.py:25,384`,
`infer_v2.py:616`).  The decode step here reads the weights once for up to `max_batch` sequences (B=4 costs 1.35x the
B=2 step, r01), so a worker that owns a GPU keeps its decode slots full: segments of any queued request are prefilled
into free slots, all slots step together, a finished slot is handed back and refilled while the others keep going
(continuous batching).  Greedy results do not depend on the company a sequence keeps -- every slot's arithmetic is
its own, in a fixed order -- which `tests/test_gpu_scheduler.py` checks against one-at-a-time decoding.
"""
import collections

import numpy as np


class Segment:
    """One text segment of one request, ready for the decode engine."""

    __slots__ = ("request", "index", "embeds", "n_left_pad", "max_new", "payload", "stream")

    def __init__(self, request, index, embeds, n_left_pad, max_new, payload=None, stream=None):
        self.request, self.index, self.embeds, self.n_left_pad, self.max_new, self.payload = request, index, embeds, n_left_pad, max_new, payload
        self.stream = stream  # beam groups: the random stream this segment draws from (None: its position in the submission order)


def live_stub(engine, like):
    """A 2-row prompt of zeros shaped like `like` [P-1, D] (what an idle slot is parked on)."""
    return like[:2] * 0


def longest_first(items, seg_of=lambda x: x):
    """Longest expected decode first (a stable sort; equal segments keep their order): the long segments start while the queue is
    full and the short ones fill the tail, instead of one long segment stepping alone at the end (64 mixed requests on 16 slots:
    71 % of the slot-steps were busy in submission order).  Expected length = the cap on new tokens, then the prompt's row count
    (codes per text token is close to constant: 11 in the bench's fixed-length mode)."""
    return sorted(items, key=lambda it: (-int(seg_of(it).max_new), -int(seg_of(it).embeds.shape[0])))


class DecodeScheduler:
    """Keeps the engine's decode slots busy with segments from a queue.

    engine: `GptEngine`-like object with prefill(slot, embeds, n_left_pad), decode(n_active, n_steps, **sampler) and
    read(slot) -> (ids, finished).  `stop_token` ends a sequence (inclusive) unless `fixed_length`.
    """

    def __init__(self, engine, max_batch, stop_token, sync_every=64):
        assert 1 <= max_batch
        self.engine, self.max_batch, self.stop_token, self.sync_every = engine, max_batch, stop_token, sync_every
        self.stats = dict(decode_calls=0, slot_steps=0, busy_slot_steps=0, refills=0)

    def run(self, segments, on_done, fixed_length=False, **sampler):
        """Decode every segment; `on_done(segment, ids)` is called as each finishes (ids: int32 array, stop token included
        when it was produced).  Order of completion is not the order of submission."""
        queue = collections.deque(longest_first(list(segments)))
        slots = [None] * self.max_batch  # (segment, steps issued so far)
        length = [0] * self.max_batch    # rows each slot's sequence holds (a freed slot below n_active keeps stepping)
        max_seq = getattr(self.engine, "max_seq", None)
        primed = 0                       # slots that have held a sequence at least once (the engine needs a prompt in every active slot)
        while queue or any(s is not None for s in slots):
            for b in range(self.max_batch):
                if slots[b] is None and queue and b <= primed:
                    seg = queue.popleft()
                    self.engine.prefill(b, seg.embeds, seg.n_left_pad)
                    if b < primed:
                        self.stats["refills"] += 1
                    primed = max(primed, b + 1)
                    slots[b] = [seg, 0]
                    length[b] = int(seg.embeds.shape[0]) + 1
            live = [b for b in range(self.max_batch) if slots[b] is not None]
            n_active = max(live) + 1
            steps = min([self.sync_every] + [slots[b][0].max_new - slots[b][1] for b in live])
            if max_seq is not None:
                for b in range(n_active):  # an idle slot under a busy one must not run off the end of its cache: restart it on a stub prompt
                    if slots[b] is None and length[b] + steps >= max_seq - 2:
                        self.engine.prefill(b, live_stub(self.engine, slots[live[0]][0].embeds), 0)
                        length[b] = 3
            if steps > 0:
                self.engine.decode(n_active, steps, suppress_stop=fixed_length, **sampler)
                self.stats["decode_calls"] += 1
                self.stats["slot_steps"] += n_active * steps
                self.stats["busy_slot_steps"] += len(live) * steps
                for b in range(n_active):
                    length[b] += steps
            for b in live:
                seg = slots[b][0]
                slots[b][1] += steps
                ids, fin = self.engine.read(b)
                if fin or slots[b][1] >= seg.max_new:
                    ids = np.asarray(ids[: seg.max_new])
                    if not fixed_length:
                        hit = np.nonzero(ids == self.stop_token)[0]
                        if hit.size:
                            ids = ids[: hit[0] + 1]
                    on_done(seg, ids)
                    slots[b] = None
        return self.stats


class BeamGroupScheduler:
    """The same idea for the served default (`num_beams=3` beam-sample, infer_v2.py:598-606): a text segment needs `num_beams`
    slots -- a beam GROUP -- and the reference decodes the segments of a request one after another (infer_v2.py:616).  Here the
    engine's groups (floor(max_batch / num_beams); group g = slots g*num_beams ..) are kept busy with queued segments: prefill
    into a free group's first slot, `beam_begin` it, step all groups together, read a group when it has finished (scorer done
    or max_new reached), refill or park it.  A segment draws from the random stream `segment.stream` (default: its position
    in the submission order), so its tokens do not depend on the group it lands in or on its company
    (tests/test_gpu_beam_groups.py holds that bit for bit against one-group-at-a-time decoding).

    engine: `GptEngine`-like with prefill(slot, embeds, pad), beam_begin(num_beams, group=, rng_stream=), beam_park(group),
    beam_decode(n_steps, groups=, **sampler), beam_read(max_new, group=) -> (ids, done, score, ...).
    """

    def __init__(self, engine, num_beams, max_groups=None, sync_every=64):
        self.engine, self.num_beams, self.sync_every = engine, int(num_beams), sync_every
        cap = engine.max_batch // self.num_beams
        if engine.max_batch <= 4:  # register GEMVs: one group
            cap = min(cap, 1)
        self.max_groups = max(1, min(cap, max_groups or cap))
        assert self.max_groups * self.num_beams <= engine.max_batch, (self.max_groups, self.num_beams, engine.max_batch)
        self.stats = dict(decode_calls=0, group_steps=0, busy_group_steps=0, refills=0)

    def run(self, segments, on_done, fixed_length=False, **sampler):
        """Decode every segment; `on_done(segment, ids, score)` as each finishes: ids = the best hypothesis as
        `generate()` returns it (`BeamSearchScorer.finalize`: + eos when there is room)."""
        # (a segment's random stream is fixed by its place in the SUBMISSION order before the queue is re-ordered: its tokens do not change)
        queue = collections.deque(longest_first([(seg, getattr(seg, "stream", None) if getattr(seg, "stream", None) is not None else i)
                                                 for i, seg in enumerate(segments)], seg_of=lambda it: it[0]))
        groups = [None] * self.max_groups  # [segment, steps issued]
        begun = 0                           # groups that have held a segment (fill from 0 upwards)
        eng, nb = self.engine, self.num_beams
        while queue or any(g is not None for g in groups):
            for g in range(self.max_groups):
                if groups[g] is None and queue and g <= begun:
                    seg, stream = queue.popleft()
                    eng.prefill(g * nb, seg.embeds, seg.n_left_pad)
                    eng.beam_begin(nb, group=g, rng_stream=stream)
                    if g < begun:
                        self.stats["refills"] += 1
                    begun = max(begun, g + 1)
                    groups[g] = [seg, 0]
            live = [g for g in range(self.max_groups) if groups[g] is not None]
            n_groups = max(live) + 1
            for g in range(n_groups):
                if groups[g] is None:
                    eng.beam_park(g)  # nothing queued for it: a no-op under the live ones
            steps = min([self.sync_every] + [groups[g][0].max_new - groups[g][1] for g in live])
            if steps > 0:
                eng.beam_decode(steps, groups=n_groups, suppress_stop=fixed_length, **sampler)
                self.stats["decode_calls"] += 1
                self.stats["group_steps"] += n_groups * steps
                self.stats["busy_group_steps"] += len(live) * steps
            for g in live:
                seg = groups[g][0]
                groups[g][1] += steps
                ids, done, score = eng.beam_read(seg.max_new, group=g)[:3]
                if done or groups[g][1] >= seg.max_new:
                    on_done(seg, np.asarray(ids), score)
                    groups[g] = None
        return self.stats
