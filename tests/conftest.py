import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
