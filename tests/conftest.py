import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running duplicate coverage (learning curves, further rehearsal shapes); skipped unless SRLX_RUN_SLOW=1 so that "
                                       "the round-end GPU suite stays far inside its time limit")


def pytest_collection_modifyitems(config, items):
    # GPU tests must fail loudly on a GPU box, but be deselected by `-m "not gpu"` here.
    # If someone runs the whole suite on a box without a GPU, skip (not pass) them.
    if os.environ.get("SRLX_RUN_SLOW", "0") != "1":
        skip_slow = pytest.mark.skip(reason="slow duplicate coverage: set SRLX_RUN_SLOW=1")
        for item in items:
            if "slow" in item.keywords:
                item.add_marker(skip_slow)
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
