import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library (it is git-ignored): build it once (hipcc cross-compiles without a GPU)
    lib = os.path.join(ROOT, "mmd_amd", "lib", "libmmd_amd.so")
    if not os.path.exists(lib) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    import parity_log
    path = parity_log.flush()
    if path:
        print(f"\nparity records written to {path}")
