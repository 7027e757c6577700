import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import orc
    orc.build()
    return orc


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first(request):
    """PyTorch's HIP initialisation fails ("No HIP GPUs are available") in a process in which the library has
    already initialised HIP, so a run that holds GPU tests lets PyTorch go first, whichever test comes first."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    yield
