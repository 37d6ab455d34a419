import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first():
    """On a GPU box let torch initialise the device before libamplihip does: a process in which HIP was first
    brought up by another library made torch's NCCL backend report "no GPUs found" (seen with ROCm 7.2 / torch 2.10)."""
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:
        pass
    yield
