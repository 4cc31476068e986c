import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: opt-in (minutes of CPU, or pure stress on the GPU); skipped unless -m names it or ZK_RUN_SLOW=1")


def pytest_collection_modifyitems(config, items):
    """`slow` tests run only when asked for: `-m slow`, `-m "gpu and slow"`, or ZK_RUN_SLOW=1.  The driver's two commands (-m "not gpu", -m gpu) skip them:
    the CPU suite stays within a few minutes and the GPU suite within its budget (tests/README.md)."""
    import os

    if "slow" in (config.getoption("-m") or "") or os.environ.get("ZK_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow: opt in with -m slow (or -m 'gpu and slow') or ZK_RUN_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


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


@pytest.fixture(scope="session")
def cpu_rsa17_proof():
    """The CPU oracle backend's proof of BASELINE configs[1] (RSA k = 17, Poseidon, witness 0): ~25 s of host time, computed ONCE per session and shared by
    the single-GPU and the two-rank byte-equality tests (tests/README.md: the GPU suite's time budget)."""
    import halo2_zkcert_amd.prover as pv
    from oracle_backend import OracleBackend

    cp = pv.Prover(OracleBackend(os.cpu_count() or 8), pv.CircuitShape.rsa(17), satisfiable=True)
    return bytes(cp.prove(cp.witness(0), transcript="poseidon")["proof"])
