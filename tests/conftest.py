import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    """Cores this process may really use (affinity mask capped by the cgroup CPU quota, as bench.usable_cores)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's CPU pool defaults to the HOST's core count; inside a CPU-quota'd container every small fp64 product of the oracle
    # then pays an oversubscribed barrier (the GPU suite: 156 CPU-minutes for 11 minutes of wall clock in round 6)
    try:
        import torch
        torch.set_num_threads(min(16, _usable_cores()))
    except Exception:
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def precision_log_path():
    """File the GPU parity tests append their measured errors to: $GCL_PRECISION_LOG if set (tools/collect_profiles.sh
    points it into gpurun_out/), else a file under the system temp directory -- never a path that may not exist on a
    clean checkout (gpurun_out/ is git-ignored)."""
    import tempfile
    path = os.environ.get("GCL_PRECISION_LOG") or os.path.join(tempfile.gettempdir(), "gcl_precision_errors.log")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    return path
