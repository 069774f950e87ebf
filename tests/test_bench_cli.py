"""bench.py host logic that needs no GPU: `--gpus N` outside torchrun starts its own ranks (a child torch.distributed.run on
127.0.0.1, before anything touches a GPU), the argument surface, and the mixed64 request list."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gpus_n_launches_its_own_ranks(monkeypatch):
    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    args = bench.parse(["--gpus", "4", "--steps", "2", "--warmup", "1"])
    assert bench.launch_ranks(args) == 7  # the child's exit code is passed on
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py" and cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_defaults_and_modes():
    import bench

    a = bench.parse([])
    assert (a.gpus, a.steps, a.warmup, a.dtype, a.workload, a.decode, a.concurrency) == (1, 3, 1, "bf16", "request", "greedy", 1)
    a = bench.parse(["--workload", "mixed64", "--slots", "8", "--decode", "beam", "--dtype", "f32"])
    assert (a.workload, a.slots, a.decode, a.dtype) == ("mixed64", 8, "beam", "f32")
    from voice_tts_amd import sharding

    reqs = sharding.mixed_requests()
    mine = [sharding.my_requests(len(reqs), r, 8) for r in range(8)]
    assert sorted(i for m in mine for i in m) == list(range(64)) and all(len(m) == 8 for m in mine)  # request i -> rank i mod 8
