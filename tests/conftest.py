import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def cpu_threads():
    """Threads for the CPU oracle legs: the box's CPU share (16 for one GPU), not the host's core count -- torch's default on the
    GPU box is 128 threads for a 16-core share, which made a 220-step oracle decode take 215 s instead of ~15."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p_))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("IXTTS_CPU_THREADS", "16"))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch

        torch.set_num_threads(cpu_threads())
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load
