import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    oracle_lib.set_threads(min(8, os.cpu_count() or 1))
    return oracle_lib


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the GPU tests.  Fails loudly (no skip, no fallback) when the
    HIP library is missing or no gfx950 device is usable."""
    from zktls_amd.device import Context
    c = Context(0)
    yield c
    c.close()
