import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """(GPU box) peak device memory of the session, next to the parity report: captured graphs are never destroyed (graph.py)"""
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "gpu_test_memory.txt"), "w") as fh:
                fh.write("reserved_GiB %.2f peak_reserved_GiB %.2f\n" % (torch.cuda.memory_reserved() / 2 ** 30,
                                                                        torch.cuda.max_memory_reserved() / 2 ** 30))
    except Exception:            # noqa: BLE001 -- diagnostics only
        pass
