"""The load-time weight distribution of bench.py --gpus N, minus the RCCL call itself: the packed arenas are exposed
as torch tensors over raw device pointers, copied (stand-in for dist.broadcast), and adopted by a second handle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_arena_copy_and_adopt_equals_loading():
    import voice_tts_amd.weights as WR
    from voice_tts_amd.pipeline import HotPath

    dev = torch.device("cuda:0")
    gcfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    bcfg = WR.tiny_bigvgan_cfg(64)
    Wg, Wb = WR.make_gpt_weights(gcfg, seed=7), WR.make_bigvgan_weights(bcfg, seed=8)
    for dtype in ("f32", "bf16"):
        a = HotPath(gcfg, bcfg, dtype=dtype, device=dev, max_batch=1, max_seq=96, max_frames=16).load(Wg, Wb)
        b = HotPath(gcfg, bcfg, dtype=dtype, device=dev, max_batch=1, max_seq=96, max_frames=16)
        src, dst = a.broadcast_tensors(), b.broadcast_tensors()
        assert len(src) == len(dst) == 5 and all(s.shape == d.shape and s.dtype == d.dtype and d.is_cuda for s, d in zip(src, dst))
        assert src[0].dtype == torch.uint8 and src[0].numel() > 1000
        for s, d in zip(src, dst):
            d.copy_(s)  # dist.broadcast(t, src=0) on the real node
        b.adopt()
        g = torch.Generator().manual_seed(1)
        emb = torch.randn(20, 128, generator=g)
        mel = (torch.randn(1, 80, 5, generator=g) * 2 - 4).clamp(-11.5, 2).to(dev)
        ia = a.generate([(emb, 0)], 12)[0]
        ib = b.generate([(emb, 0)], 12)[0]
        assert ia.tolist() == ib.tolist()
        assert torch.equal(a.bigvgan(mel), b.bigvgan(mel))


def test_rccl_broadcast_of_raw_arena_tensors_single_rank():
    """The RCCL call itself, on a one-rank group (the box has one GPU): `dist.broadcast` accepts the tensors that wrap the
    library's raw device arenas, and the collectives bench.py uses (barrier, all_reduce MAX) run on them."""
    import socket

    import torch.distributed as dist

    import voice_tts_amd.weights as WR
    from voice_tts_amd.pipeline import HotPath

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        gcfg, bcfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2), WR.tiny_bigvgan_cfg(64)
        hp = HotPath(gcfg, bcfg, dtype="bf16", device=dev, max_batch=1, max_seq=96, max_frames=16).load(
            WR.make_gpt_weights(gcfg, seed=7), WR.make_bigvgan_weights(bcfg, seed=8))
        before = [t.clone() for t in hp.broadcast_tensors()]
        for t in hp.broadcast_tensors():
            dist.broadcast(t, src=0)
        dist.barrier()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(before, hp.broadcast_tensors()))
        t = torch.tensor([1.5], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()
