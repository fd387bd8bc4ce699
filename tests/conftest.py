import os
import sys

# tests/test_gpu_tp.py runs the ranks of a tensor-parallel group as handles of ONE process, each with its compute stream and
# the library's exchange stream, and a rank's wait kernel spins until its peer's launches run: every stream needs a hardware
# queue of its own (the HIP runtime multiplexes streams over 4 by default; read when the runtime initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
