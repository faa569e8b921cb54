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
    try:
        from craytracer_amd import api
        return api.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    import oraclelib
    have_ref = oraclelib.have_ref()
    for item in items:
        if "ref" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="oracle/_ref not built (only possible where /root/reference exists)"))
