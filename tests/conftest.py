import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: builds the mixed-phase oracle tables (tens of seconds, cached)")


@pytest.fixture(scope="session")
def oracle_warm():
    from oracle.oracle import Oracle
    o = Oracle(iiwarm=True)
    yield o
    o.close()


@pytest.fixture(scope="session")
def oracle_mixed():
    from oracle.oracle import Oracle
    o = Oracle(iiwarm=False)        # tables cached under oracle/_cache after the first build
    yield o
    o.close()


@pytest.fixture(scope="session")
def gpu_mixed():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible; the HIP path has no CPU fallback")
    from kid_amd import ThompsonMP
    m = ThompsonMP(iiwarm=False)
    yield m
    m.close()


@pytest.fixture(scope="session")
def gpu_warm():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible; the HIP path has no CPU fallback")
    from kid_amd import ThompsonMP
    m = ThompsonMP(iiwarm=True)
    yield m
    m.close()


@pytest.fixture(scope="session")
def oracle_mixed_aero():
    from oracle.oracle import Oracle
    o = Oracle(iiwarm=False, aerosol_aware=True)
    yield o
    o.close()


@pytest.fixture(scope="session")
def gpu_mixed_aero():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible; the HIP path has no CPU fallback")
    from kid_amd import ThompsonMP
    m = ThompsonMP(iiwarm=False, aerosol_aware=True)
    yield m
    m.close()
