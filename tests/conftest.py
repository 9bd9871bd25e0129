import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.lib()  # builds liboracle_cpu.so on first use (gcc only)
    return o


@pytest.fixture(scope="session")
def mli():
    """The product library.  No fallback: if it is not built the GPU tests must fail, not skip."""
    from min_llm_inference_amd import load_library
    return load_library()


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")
