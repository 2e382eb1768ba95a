"""Device arithmetic of bmm-mcmc_amd/csrc/bmm_spec.h, evaluated on the GPU through the
C ABI, against the oracle's plain-C restatement: every result must be bit-identical,
because the categorical draw of z_n is decided by comparisons of these values."""
import ctypes as C

import numpy as np
import pytest

from bmm_mcmc_amd import _capi

pytestmark = pytest.mark.gpu


def _dev(op, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    y2 = None if y is None else np.ascontiguousarray(y, dtype=np.float64)
    _capi.check(_capi.lib().bmm_device_math(0, op, _capi.vp(x), None if y2 is None else _capi.vp(y2),
                                            _capi.vp(out), C.c_int64(x.size)))
    return out


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def test_log_bit_exact(oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 400000)), rng.uniform(0.5, 2.0, 400000),
                        0.5 + rng.integers(0, 10 ** 7, 400000), [0.0, 1.0, 5e-324, 1e-310, np.inf]])
    assert np.array_equal(_bits(_dev(0, x)), _bits(oracle.log_array(x)))


def test_exp_bit_exact(oracle):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-720, 0, 600000), rng.uniform(-30, 30, 300000),
                        rng.uniform(0, 709, 100000), [0.0, -708.0, -708.0000001, -np.inf, 710.0]])
    assert np.array_equal(_bits(_dev(1, x)), _bits(oracle.exp_array(x)))


def test_expw_bit_exact_including_subnormal_results(oracle):
    """The draw's weight exponential: v_ldexp_f64 must round results below 2^-1022 exactly as the
    host's ldexp does (the spec leaves that range to IEEE rounding instead of flushing by comparison)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-760, 0, 800000), rng.uniform(-745.5, -707.5, 300000), rng.uniform(-40, 0, 300000),
                        -np.exp(rng.uniform(-40, 3, 100000)),
                        [0.0, -0.0, -708.0, -745.13, -745.14, -746.0, -1000.0, -1e300, -np.inf, np.nan]])
    got, want = _dev(4, x), oracle.expw_array(x)
    assert np.array_equal(_bits(got), _bits(want))
    assert got[-1] == 0.0 and got[-2] == 0.0 and got[0 - 10] == 1.0


def test_division_and_sqrt_are_correctly_rounded():
    rng = np.random.default_rng(3)
    a = np.exp(rng.uniform(-300, 300, 1000000)) * rng.choice([-1.0, 1.0], 1000000)
    b = np.exp(rng.uniform(-300, 300, 1000000))
    assert np.array_equal(_bits(_dev(2, a, b)), _bits(a / b))
    s = np.exp(rng.uniform(-700, 700, 1000000))
    assert np.array_equal(_bits(_dev(3, s)), _bits(np.sqrt(s)))


@pytest.mark.parametrize("shape", [0.5, 1.0, 3.7, 250.5, 1e6])
def test_gamma_variates_bit_exact(oracle, shape):
    n = 20000
    out = np.empty(n)
    _capi.check(_capi.lib().bmm_device_variates(0, 0, C.c_double(shape), C.c_double(0), C.c_uint64(77),
                                                C.c_uint32(5), _capi.vp(out), C.c_int64(n)))
    want = np.array([oracle.rgamma(shape, 77, i, 5, 3) for i in range(n)])
    assert np.array_equal(_bits(out), _bits(want))


def test_beta_and_alpha_variates_bit_exact(oracle):
    n = 20000
    out = np.empty(n)
    L = _capi.lib()
    _capi.check(L.bmm_device_variates(0, 1, C.c_double(0.5), C.c_double(120.5), C.c_uint64(9), C.c_uint32(2),
                                      _capi.vp(out), C.c_int64(n)))
    want = np.array([oracle.rbeta(0.5, 120.5, 9, i, i, 2, 3, 4) for i in range(n)])
    assert np.array_equal(_bits(out), _bits(want))
    n = 5000
    out = np.empty(n)
    _capi.check(L.bmm_device_variates(0, 2, C.c_double(1.3), C.c_double(4), C.c_uint64(100), C.c_uint32(8),
                                      _capi.vp(out), C.c_int64(n)))
    want = np.array([oracle.update_alpha(1.3, 1.0, 1.0, 1000, 4, 100 + i, 8) for i in range(n)])
    assert np.array_equal(_bits(out), _bits(want))
