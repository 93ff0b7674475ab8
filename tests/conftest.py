import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
