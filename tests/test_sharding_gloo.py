"""N>1 path on CPU: world_size-2 gloo run of the load-time broadcast and the request sharding."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from voice_tts_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # packed "arenas": rank 0 holds the weights, the others hold garbage until the broadcast
    g = torch.Generator().manual_seed(3)
    arenas = [torch.randint(0, 255, (4096,), dtype=torch.uint8, generator=g), torch.randn(300, generator=g)]
    if rank != 0:
        arenas = [torch.zeros_like(a) for a in arenas]
    sharding.broadcast_weights(arenas, src=0)
    # glue weights: shapes known everywhere, tensors only on rank 0 -> one packed message
    shapes = [("a.weight", (3, 5)), ("a.bias", (5,)), ("b", (2, 2, 2))]
    W = {n: torch.randn(s, generator=g) for n, s in shapes} if rank == 0 else None
    got = sharding.broadcast_tensor_dict(W, shapes, "cpu", src=0)
    assert [tuple(got[n].shape) for n, _ in shapes] == [s for _, s in shapes]
    glue_sum = float(sum(v.double().sum() for v in got.values()))
    # a checkpoint's tensors, names and shapes known to rank 0 only (IndexTTS2(weight_broadcast=...)): metadata first, one packed buffer after
    sd = {"gpt/text_embedding.weight": torch.randn(7, 4, generator=g), "s2mel/cfm.x": torch.randn(3, generator=g), "emo_matrix": torch.randn(5, 2, generator=g)} if rank == 0 else None
    got2, meta = sharding.broadcast_state_dict(sd, "cpu", src=0, meta=dict(bcfg={"upsample_initial_channel": 64}, present=["gpt", "s2mel", "emo_matrix"]) if rank == 0 else None)
    assert meta == dict(bcfg={"upsample_initial_channel": 64}, present=["gpt", "s2mel", "emo_matrix"])
    assert {k: tuple(v.shape) for k, v in got2.items()} == {"gpt/text_embedding.weight": (7, 4), "s2mel/cfm.x": (3,), "emo_matrix": (5, 2)}
    glue_sum += float(sum(v.double().sum() for v in got2.values()))
    mine = sharding.my_requests(7, rank, world)
    audio, elapsed = sharding.gather_throughput(10.0 * len(mine), 1.0 + rank)
    q.put((rank, [int(a.sum()) if a.dtype == torch.uint8 else float(a.sum()) for a in arenas] + [glue_sum], mine, audio, elapsed))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_round_robin_world2():
    from voice_tts_amd import sharding

    assert sharding.assign(7, 2) == [[0, 2, 4, 6], [1, 3, 5]]
    assert sharding.assign(3, 8) == [[0], [1], [2], [], [], [], [], []]
    reqs = sharding.mixed_requests()
    assert len(reqs) == 64 and all(50 <= sum(r) <= 400 and max(r) <= 120 and max(r) - min(r) <= 1 for r in reqs)
    assert [len(r) for r in reqs] == [-(-sum(r) // 120) for r in reqs] and reqs == sharding.mixed_requests()  # seeded
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, sums0, mine0, audio0, el0), (r1, sums1, mine1, audio1, el1) = res
    assert sums0 == sums1  # every rank ends with rank 0's weights
    assert mine0 == [0, 2, 4, 6] and mine1 == [1, 3, 5]
    assert audio0 == audio1 == 70.0 and el0 == el1 == 2.0  # sum of units, max of times
