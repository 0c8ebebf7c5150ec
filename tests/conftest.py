import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device: on a machine without one they are skipped, not failed (the product path itself
    still fails loudly there, tests/test_cabi_cpu.py)."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device (torch.cuda.is_available() is False)")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
