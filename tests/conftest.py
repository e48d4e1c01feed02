import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sglang-fluentllm_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _gpu_tests_do_not_overlap(request):
    """A GPU test starts on an idle device and leaves one: device work left in flight on a side stream (a graph replay, a second-stream launch) must
    not be writing into blocks the caching allocator hands to the NEXT test.  (Hygiene: it did not cure round 6's soak failure — DESIGN.md section 7
    item 5 — whose cause lay elsewhere.  No effect on CPU tests: torch.cuda is not touched without the `gpu` marker.)"""
    is_gpu = request.node.get_closest_marker("gpu") is not None
    if is_gpu:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    yield
    if is_gpu:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
