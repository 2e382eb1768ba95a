import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch bundles its own HIP runtime; it must be the first libamdhip64 in the process (as in
    # bench.py), or torch later binds to the system one and reports no GPUs.  Tests that hand torch
    # device tensors to the library (sharded chains) need both to share one runtime.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
