"""Request sharding across GPUs: independent units, zero exchange.

The reference runs one gunicorn worker per GPU and lets the kernel's accept() pick a worker
(`server.py:485-551`, `gunicorn_config.py:43-60`: GPU = gpus[(worker.age-1) % len(gpus)]).
The same policy, explicit: request i -> rank i mod N; every rank holds a full weight copy,
broadcast once at load from rank 0 (RCCL when the tensors are on GPUs, any backend otherwise).
"""
import torch


def assign(n_requests, world_size):
    """[[request ids of rank 0], [rank 1], ...] -- round-robin, order-preserving."""
    return [list(range(r, n_requests, world_size)) for r in range(world_size)]


def my_requests(n_requests, rank, world_size):
    return list(range(rank, n_requests, world_size))


def broadcast_weights(tensors, src=0, group=None):
    """One broadcast per packed arena / table (a handful of large messages, not per-tensor chatter)."""
    import torch.distributed as dist

    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def gather_throughput(local_audio_seconds, local_elapsed, device=None, group=None):
    """(sum of audio seconds over ranks, max elapsed over ranks) -- the aggregate rate is their quotient."""
    import torch.distributed as dist

    a = torch.tensor([float(local_audio_seconds)], dtype=torch.float64, device=device)
    e = torch.tensor([float(local_elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(a, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(e, op=dist.ReduceOp.MAX, group=group)
    return float(a.item()), float(e.item())
