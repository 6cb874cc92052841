import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` need a HIP device (there is no CPU path to fall back to): on a machine without one they are skipped
    instead of failing with 'No HIP GPUs', so a plain `pytest tests/` is green on CPU-only CI."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X); none visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The C oracle is test infrastructure; (re)build it if its source is newer than the .so files."""
    from oracle import oracle as _o
    _o.build()
