import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """-m gpu tests must fail loudly (not skip) when the GPU or the HIP library is missing on a GPU box;
    on a CPU-only box they are simply deselected by `-m "not gpu"`."""
    return


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)
