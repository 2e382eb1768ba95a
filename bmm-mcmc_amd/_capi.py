"""ctypes binding of include/bmm_mcmc.h.  There is no fallback: if the HIP library is
missing or no GPU is visible, calls raise."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None

SAMPLER_COLLAPSED, SAMPLER_DP, SAMPLER_SB, SAMPLER_FULL = 0, 1, 2, 3
SAMPLER_CODE = {"collapsed": 0, "dp": 1, "stickbreaking": 2, "full": 3}
NA_INTEGER = -2147483648

# every symbol include/bmm_mcmc.h declares
SYMBOLS = [
    "bmm_last_error", "bmm_spec_group_width", "bmm_spec_group_width_own", "bmm_spec_group_width_for", "bmm_default_batch", "bmm_collapsed_run", "bmm_dp_run",
    "bmm_sb_run", "bmm_full_run", "bmm_chain_create", "bmm_chain_destroy", "bmm_chain_set_data_host",
    "bmm_chain_set_data_device", "bmm_chain_set_x_layout", "bmm_chain_get_x_layout", "bmm_chain_set_initial_labels", "bmm_chain_set_initial_params",
    "bmm_chain_sweeps", "bmm_chain_sweeps_counts", "bmm_chain_sweep_probs", "bmm_chain_set_shard", "bmm_chain_shard_resample",
    "bmm_chain_shard_deltas", "bmm_chain_shard_finish", "bmm_chain_sync", "bmm_chain_sweep_index", "bmm_chain_get_labels",
    "bmm_chain_get_counts", "bmm_chain_get_alpha", "bmm_chain_get_params", "bmm_chain_profile",
    "bmm_chain_profile_read", "bmm_chain_kernel_shape", "bmm_chain_kernel_form", "bmm_chain_batch", "bmm_device_math", "bmm_device_variates",
    "bmm_device_count", "bmm_collapsed_run_probs", "bmm_dp_run_probs", "bmm_sb_run_probs", "bmm_full_run_probs",
    "bmm_multi_run", "bmm_multi_selfcheck", "bmm_chains_sweeps", "bmm_chain_share_data", "bmm_chain_planes",
    "bmm_chain_planes_filled", "bmm_chain_shard_resample_async", "bmm_chain_stream",
    "bmm_chains_broadcast_planes", "bmm_set_progress", "bmm_last_run_phases", "bmm_host_threads", "bmm_multi_plan", "bmm_release_pools",
]


class BmmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def lib_path():
    """The product library, unless BMM_LIB_PATH names another build of it (the -DBMM_DEBUG_HOOKS test
    variant, a diagnostic or experimental build): alternatives are loaded from their own path, the
    product file is never overwritten."""
    return os.environ.get("BMM_LIB_PATH") or _build.LIB


def load(path):
    """dlopen one build of the library and declare the non-int signatures."""
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950). This package has no CPU fallback.")
    L = C.CDLL(path)
    L.bmm_last_error.restype = C.c_char_p
    L.bmm_default_batch.restype = C.c_int64
    L.bmm_default_batch.argtypes = [C.c_int, C.c_int64]
    L.bmm_chain_batch.restype = C.c_int64
    L.bmm_chain_batch.argtypes = [C.c_void_p]
    L.bmm_chain_destroy.restype = None
    L.bmm_chain_destroy.argtypes = [C.c_void_p]
    L.bmm_chain_share_data.argtypes = [C.c_void_p, C.c_void_p]
    return L


def lib():
    global _LIB
    if _LIB is None:
        _LIB = load(lib_path())
    return _LIB


def check(rc):
    if rc != 0:
        raise BmmError(rc, lib().bmm_last_error().decode())


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    lib().bmm_device_count(C.byref(n))
    return n.value


def as_x(data):
    """N x P integer matrix in R's layout (column-major int32).  That the values are 0/1 is checked by the
    library when the matrix is handed over (in the same pass that packs it): a second pass here would cost
    as much as 100 sweeps at the north-star shape."""
    X = np.asarray(data)
    if X.ndim != 2:
        raise ValueError("data must be a matrix with observations in rows")
    if X.dtype.kind == "f":
        if not np.all(X == np.round(X)):
            raise ValueError("data must be integer valued")
    elif X.dtype.kind not in "iub":
        raise ValueError("data must be an integer (0/1) matrix")
    if X.dtype != np.int32 and X.size and (X.min() < 0 or X.max() > 1):
        raise ValueError("data must be binary (0/1)")  # before a narrowing cast could hide it
    return np.asfortranarray(X, dtype=np.int32)
