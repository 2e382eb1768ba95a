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


@pytest.fixture
def dbg_lib(monkeypatch):
    """The -DBMM_DEBUG_HOOKS variant of the library (lib/libbmmmcmc_hip_dbg.so) in place of the product for
    one test: only it reads the kernel-steering switches BMM_DEBUG_GENERIC / BMM_DEBUG_THREADS /
    BMM_DEBUG_NOSPLIT / BMM_X_LAYOUT_INT32, through which the parity tests reach every kernel variant."""
    from bmm_mcmc_amd import _capi, build
    if build.stale(build.LIB_DBG):
        build.build(debug_variant=True)
    monkeypatch.setattr(_capi, "_LIB", _capi.load(build.LIB_DBG))
    return monkeypatch
