import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Load order matters in a process that uses both PyTorch-ROCm and the product library: torch ships its own HIP/HSA runtime
# and must be imported BEFORE lib/libcilqr_hip.so pulls in /opt/rocm's, or torch later reports "No HIP GPUs are available"
# (seen when only test_gpu_parity.py was selected and torch was first touched in the middle of the run).
try:
    import torch  # noqa: F401,E402
except ImportError:  # CPU-only checks that never touch torch still run
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.build(ref=True)
    return O


@pytest.fixture(scope="session")
def cilqr():
    """The product's Python host binding over the C-ABI; builds lib/libcilqr_hip.so if it is missing."""
    import cilqr_amd
    if not os.path.exists(cilqr_amd.LIB_PATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return cilqr_amd


def load_golden(name):
    import json
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)
