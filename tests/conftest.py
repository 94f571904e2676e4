import os
import sys
import warnings

import pytest

# The configuration the docs recommend for every multi-slot user (README, INTEGRATION.md, bench.py): four hardware queues, set BEFORE
# torch initialises HIP (the runtime reads it once).  PipelinedValidation warns when its slots outnumber the configured queues;
# with this the pipelined tests run in the documented configuration and the suite is warning-free.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

warnings.filterwarnings("ignore", message=".*nested tensors.*")
warnings.filterwarnings("ignore", message=".*enable_nested_tensor.*")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with -m gpu on the GPU box")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
