import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def usable_cpus() -> int:
    """CPUs this process may really use: min(os.cpu_count, affinity, cgroup quota).  The GPU box shows 128 cores but
    grants a job a 16-CPU share: torch-CPU oracles started with 128 OpenMP threads there spin against the quota and
    run orders of magnitude slower."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_sessionstart(session):
    try:
        import torch
        torch.set_num_threads(max(1, min(8, usable_cpus())))     # the oracle's nets are tiny: 8 threads is plenty
    except ImportError:
        pass
