import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs small torch ops: a 256-thread intra-op pool (the GPU box's default) makes them far slower
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def hip_library_is_current():
    """Build ppde_amd/libppde_hip.so if it is missing or older than its sources (hipcc cross-compiles without a GPU).
    The product path itself never builds or falls back: it raises when the library is missing."""
    from ppde_amd import build
    try:
        build.build()
    except (FileNotFoundError, OSError):
        pass                        # no hipcc here: the tests that need the library will say so
    yield
