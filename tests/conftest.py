import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _cpu_budget() -> int:
    """CPUs this process may really use: affinity mask and cgroup quota, capped at 64.  The oracles are OpenMP / torch-CPU code; on a
    shared box the default (one thread per logical CPU of the host) oversubscribes the container's share and runs many times slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_budget()))
# two OpenMP runtimes live in this process (the oracle's libgomp and torch's bundled copy): bound how long an idle pool spins
# after a parallel region, so that it does not eat the other one's share of a small CPU quota
os.environ.setdefault("GOMP_SPINCOUNT", "30000")


def pytest_configure(config):
    try:
        import torch
        torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes tens of seconds (full-size configurations); collected last")


def pytest_collection_modifyitems(config, items):
    """Cheap tests first: every hot-path row's oracle / fixture tests run before the full-size cases (marked `slow`) and the
    bench.py subprocess, so a slow tail can never hide a row's parity result behind a time limit.  Stable within a class."""
    def rank(item):
        if "test_gpu_bench_contract" in item.nodeid:
            return 2
        return 1 if item.get_closest_marker("slow") else 0
    items.sort(key=rank)


@pytest.fixture(autouse=True)
def _restore_cpu_threads():
    """ModelWorker pins torch (and with it the process's OpenMP default) to ONE thread — right for a serving process, but every
    oracle call of a later test then runs single-threaded (the full-width g21 cases: 40 s -> 270 s).  Put the budget back per test."""
    n = int(os.environ["OMP_NUM_THREADS"])
    try:
        import torch
        if torch.get_num_threads() != n:
            torch.set_num_threads(n)
    except Exception:
        pass
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(n)
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]
    return load


def bf16_close(a_bits, b_bits, ulps=2, atol=0.0):
    """|a-b| <= ulps * bf16-ulp(max(|a|,|b|)) + atol, elementwise, on bf16 bit patterns."""
    from oracle import voxref as vr
    a, b = vr.bf2f(a_bits).astype(np.float64), vr.bf2f(b_bits).astype(np.float64)
    mag = np.maximum(np.abs(a), np.abs(b))
    ulp = np.where(mag > 0, 2.0 ** (np.floor(np.log2(np.maximum(mag, 1e-38))) - 7), 0.0)
    return np.abs(a - b) <= ulps * ulp + atol
