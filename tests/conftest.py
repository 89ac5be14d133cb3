import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def pytest_collection_modifyitems(config, items):
    """Plain `pytest` on a box without a GPU skips the gpu-marked tests instead of erroring in them."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if gpu_items and _gpu_count() == 0:
        skip = pytest.mark.skip(reason="no HIP device visible (hipGetDeviceCount() == 0)")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure): built with g++ on first use."""
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def opt_lib():
    """libOpt.so, the HIP product library.  Built in-tree if missing (hipcc cross-compiles on CPU boxes)."""
    from opt_amd import api, build
    if not os.path.exists(api.LIB_PATH):
        build.build()
    return api
