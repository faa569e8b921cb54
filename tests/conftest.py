import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference's own code; build container only)")


def _has_gpu():
    """False only when the library loads and reports no device.  A missing or broken libcrt_gpu.so is NOT
    a reason to skip: the gpu tests then run and fail loudly (there is no CPU fallback to hide behind)."""
    try:
        from craytracer_amd import api
        return api.device_count() > 0
    except Exception:
        return True


def pytest_collection_modifyitems(config, items):
    import oraclelib
    have_ref = oraclelib.have_ref()
    have_gpu = None                                # asked once, and only when a gpu test was collected
    for item in items:
        if "ref" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="oracle/_ref not built (only possible where /root/reference exists)"))
        if "gpu" in item.keywords:
            if have_gpu is None:
                have_gpu = _has_gpu()
            if not have_gpu:
                item.add_marker(pytest.mark.skip(reason="no MI355X visible (the traversal path has no CPU fallback)"))
