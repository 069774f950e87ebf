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


def broadcast_tensor_dict(W, shapes, device, src=0, group=None, rank=None):
    """Glue weights as ONE message: every rank knows `shapes` = [(name, shape)] (they follow from the config), rank `src` holds
    the tensors `W`; they are packed into a single flat fp32 buffer on `device`, broadcast once, and handed back as views of that
    buffer -- the other ranks pass W=None and never generate or read the weights themselves."""
    import math

    import torch.distributed as dist

    rank = dist.get_rank(group) if rank is None else rank
    sizes = [int(math.prod(shape)) for _, shape in shapes]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    if rank == src:
        o = 0
        for (name, shape), n in zip(shapes, sizes):
            assert tuple(W[name].shape) == tuple(shape), (name, tuple(W[name].shape), tuple(shape))
            flat[o:o + n].copy_(W[name].reshape(-1))
            o += n
    dist.broadcast(flat, src=src, group=group)
    out, o = {}, 0
    for (name, shape), n in zip(shapes, sizes):
        out[name] = flat[o:o + n].view(shape)
        o += n
    return out


def broadcast_state_dict(sd, device, src=0, group=None, meta=None):
    """A dict of float tensors whose NAMES AND SHAPES only rank `src` knows (it read them from a checkpoint): the names, shapes
    and `meta` (any small picklable object of our own making) go first through `broadcast_object_list`, then every tensor in
    ONE flat fp32 buffer on `device`.  The other ranks pass sd=None.  Returns ({name: view into the buffer}, meta)."""
    import math

    import torch.distributed as dist

    rank = dist.get_rank(group)
    head = [None]
    if rank == src:
        head = [([(k, tuple(v.shape)) for k, v in sd.items()], meta)]
    dist.broadcast_object_list(head, src=src, group=group)
    shapes, meta = head[0]
    return broadcast_tensor_dict(sd if rank == src else None, shapes, device, src=src, group=group, rank=rank), meta


def mixed_requests(n_requests=64, lo=50, hi=400, seed=5, max_tokens_per_segment=120, codes_per_token=11):
    """BASELINE configs[3] / SURVEY 8(d) config 4: `n_requests` texts of randint(lo, hi+1) characters (seed 5), one token per
    character, split into ceil(len / 120) near-equal segments (what `split_segments` yields for unpunctuated text), 11 mel
    codes per token.  -> [[tokens per segment of request 0], ...]"""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (n_requests,), generator=g).tolist()
    out = []
    for n in lens:
        k = -(-n // max_tokens_per_segment)
        out.append([n // k + (1 if i < n % k else 0) for i in range(k)])
    return out
