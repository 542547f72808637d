import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the compiled reference; only where /root/reference exists)")
    expr = config.getoption("markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr:
        # One HIP runtime per process: torch ships its own libamdhip64 and reports "No HIP GPUs" when /opt/rocm's copy (which
        # libmeshclust2_hip.so would pull in) is loaded first. The tests that hand device memory to torch (RCCL views) need
        # torch imported before the library, exactly as bench.py does.
        import torch  # noqa: F401


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def ref():
    from oracle import ref_py
    if not ref_py.available():
        pytest.skip("oracle/_ref/libmsc_ref.so not built (needs /root/reference)")
    ref_py.lib()
    return ref_py


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
