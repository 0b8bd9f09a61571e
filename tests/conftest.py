import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# PyTorch's ROCm wheel carries its own copy of the HIP runtime and loads it by path: a process that has already
# initialised /opt/rocm's runtime through libcornetto_hip.so then finds "no ROCm-capable device" in torch.  The tests that
# use torch for device memory import it lazily, so fix the order here: torch's runtime first, the library binds to it.
try:
    import torch  # noqa: F401
except Exception:      # CPU-only tests do not need it
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def dacc():
    """a handle of the DEVELOPMENT build of the library (libcornetto_hip_dev.so: the same sources with -DCN_DEV), the only build that reads
    the development switches (CORNETTO_SDUST_CHUNK, CORNETTO_SDUST_SIFT, CORNETTO_SIFT_DP, *_EST_FORCE ...) from the environment; the tests
    that force decompositions, kernel families and failing estimates use it, everything else runs on the product build"""
    import cornetto_amd
    a = cornetto_amd.Accel(0, dev=True)
    yield a
    a.close()
