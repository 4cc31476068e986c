import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import zkoracle_py as zo

    zo.build()
    zo.lib()
    return zo


@pytest.fixture(scope="session")
def zk():
    """The product binding + one GPU context (GPU tests only; fails loudly without the HIP library)."""
    import halo2_zkcert_amd.ffi as ffi

    ffi.lib()
    ctx = ffi.Context(0)
    yield ffi, ctx
    ctx.close()
