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


def parity_record(name: str, numbers: dict) -> None:
    """Parity numbers a GPU test prints (graphs with a differing top-k mask, max |logit diff|, ...) also go on record: one JSON
    object per test name in gpurun_out/parity_report.json (merged back from the GPU box by gpurun; beside the test log of whoever
    runs the suite from the repository root) -- the counts are bounded by asserts, the report says what they WERE."""
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_report.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = numbers
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass                # a read-only checkout: the printed line stays
    print(f"[parity] {name}: {json.dumps(numbers)}")
