import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    config.addinivalue_line("markers", "slow: long GPU cases (`-m gpu` still selects them); collected LAST so that a step "
                                       "limit can only cost them")
    # The torch-CPU oracle is the checker of most GPU tests.  On a 2 x 64-core GPU host torch's default thread count
    # (all logical CPUs) is 3x SLOWER for these small 1-D convolutions than 32 threads (bench.py cpu_baseline.legs).
    # A GPU box gives one GPU's share of the host (16 CPUs) to a job whatever os.cpu_count() says, so: min(16, affinity,
    # cgroup quota).
    try:
        import torch

        torch.set_num_threads(cpu_share())
    except ImportError:
        pass


def cpu_share(cap=16):
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:  # cgroup v2 quota: "max 100000" or "<quota> <period>"
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(cap, n))


def pytest_collection_modifyitems(config, items):
    """Stable order, `slow` cases at the end."""
    items.sort(key=lambda it: 1 if it.get_closest_marker("slow") else 0)


@pytest.fixture(scope="session")
def lib():
    from volpick_amd import _lib

    if not _lib.LIB_PATH.exists():
        import __graft_entry__ as g

        g.build()
    return _lib.load()
