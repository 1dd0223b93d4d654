import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
sys.path.insert(0, ROOT)


def _usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (bench.py usable_cpus)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            n = max(1, min(n, int(float(q) / float(period))))
    except Exception:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, int(q / period)))
        except Exception:
            pass
    return n


# The CPU oracle (the checker of the GPU tests) runs its instances under OpenMP.  The GPU box shows 256 hardware threads to a
# container whose cgroup grants 16 CPU-seconds per second: 256 spinning threads there spend most of the quota waiting for
# each other (the same suite took 426 s on one box and 596 s on the next, 92 vs 135 CPU-minutes).  One thread per usable CPU,
# sleeping when idle -- set before anything loads an OpenMP runtime; the workers of the subprocess tests inherit it.
os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cpus()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the long variant of a lock-step run whose shorter form is in the suite as well (MEMGYM_FAST=1 leaves them out)")


def pytest_collection_modifyitems(config, items):
    """Tests marked `slow` are the long variants of lock-step runs whose shorter form is in the suite as well (rare paths
    need long runs).  They run by default -- with the oracle on one OpenMP thread per usable CPU the whole GPU suite takes ~5
    minutes -- and MEMGYM_FAST=1 leaves them out."""
    import pytest

    if not os.environ.get("MEMGYM_FAST"):
        return
    skip = pytest.mark.skip(reason="slow variant: unset MEMGYM_FAST")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)
