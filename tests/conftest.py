import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first(request):
    """torch bundles its own libamdhip64.so, libmpcmp.so links the system one (/opt/rocm): two HIP runtimes in one process.  When the
    system runtime touches the GPU first, torch's later initialisation fails with "No HIP GPUs are available" (seen when the test files
    run in another order than the alphabetical one); with torch's runtime first both work — the order bench.py has.  So: when GPU tests are
    selected, let torch initialise the device before any test creates an mpcmp context."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.zeros(1, device="cuda")
        except Exception:
            pass
    yield
