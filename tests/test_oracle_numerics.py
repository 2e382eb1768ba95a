"""The oracle's restated numerics: Philox known answers, log/exp accuracy, variates."""
import numpy as np


def test_philox_random123_kats(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert oracle.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    f = 0xffffffff
    assert oracle.philox4x32_10((f, f, f, f), (f, f)) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert oracle.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
                                (0xa4093822, 0x299f31d0)) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_philox2x32_random123_kats(oracle):
    # Random123 kat_vectors, philox2x32 10 rounds: the generator of the per-observation uniform
    assert oracle.philox2x32_10((0, 0), 0) == (0xff1dae59, 0x6cd10df2)
    f = 0xffffffff
    assert oracle.philox2x32_10((f, f), f) == (0x2c3f628b, 0xab4fd7ad)
    assert oracle.philox2x32_10((0x243f6a88, 0x85a308d3), 0x13198a2e) == (0xdd7ce038, 0xf62a4c12)


def test_uniform_range_and_resolution(oracle):
    L = oracle.lib()
    assert L.oracle_u01(0, 0) == 0.0
    assert L.oracle_u01(0xffffffff, 0xffffffff) == 1.0 - 2.0 ** -53
    # the draw's uniform: 52 bits, largest value 1 - 2^-52, so that u * t < t for every finite t > 0
    assert L.oracle_u52(0, 0) == 0.0
    assert L.oracle_u52(0xffffffff, 0xffffffff) == 1.0 - 2.0 ** -52
    assert L.oracle_u52(0x80000000, 0) == 0.5
    assert L.oracle_u52(0, 0x1000) == 2.0 ** -52 and L.oracle_u52(0, 0xfff) == 0.0
    umax = 1.0 - 2.0 ** -52
    t = np.concatenate([np.random.default_rng(0).uniform(1.0, 64.0, 100000), [1.0, 2.0, 1.0 + 2.0 ** -52, 20.0]])
    assert (umax * t < t).all()
    u = np.array([oracle.z_uniform(7, i, 3) for i in range(20000)])
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    # different sweep / seed / index give different draws
    assert oracle.z_uniform(7, 5, 3) != oracle.z_uniform(7, 5, 4)
    assert oracle.z_uniform(7, 5, 3) != oracle.z_uniform(8, 5, 3)
    assert oracle.z_uniform(7, 5 + 2 ** 32, 3) != oracle.z_uniform(7, 5, 3)


def _ulps(a, b):
    return np.abs(a - b) / np.spacing(np.abs(b))


def test_log_within_one_ulp_of_libm(oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 200000)), rng.uniform(0.5, 2.0, 200000),
                        0.5 + rng.integers(0, 10 ** 7, 200000), [1.0, 0.5, 2.0, 5e-324, 1e-310, 1.7e308]])
    y = oracle.log_array(x)
    ref = np.log(x)
    ok = ref != 0
    assert _ulps(y[ok], ref[ok]).max() <= 1.0
    assert oracle.lib().oracle_log(1.0) == 0.0
    assert oracle.lib().oracle_log(0.0) == -np.inf
    assert np.isnan(oracle.lib().oracle_log(-1.0))
    assert oracle.lib().oracle_log(np.inf) == np.inf


def test_exp_within_one_ulp_of_libm_and_flush(oracle):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-708, 0, 300000), rng.uniform(-30, 30, 200000), rng.uniform(0, 709, 50000)])
    y = oracle.exp_array(x)
    assert _ulps(y, np.exp(x)).max() <= 1.0
    L = oracle.lib()
    assert L.oracle_exp(0.0) == 1.0
    assert L.oracle_exp(-708.5) == 0.0 and L.oracle_exp(-np.inf) == 0.0
    assert L.oracle_exp(710.0) == np.inf
    assert np.isnan(L.oracle_exp(np.nan))


def test_expw_within_one_ulp_and_underflow(oracle):
    """The draw's weight exponential (x <= 0): < 1 ulp of libm in the normal range, correctly
    rounded subnormals by construction (ldexp), exact zeros below and for -inf / NaN."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-708, 0, 400000), rng.uniform(-40, 0, 200000), -np.exp(rng.uniform(-40, 3, 100000))])
    y = oracle.expw_array(x)
    assert _ulps(y, np.exp(x)).max() <= 1.0
    L = oracle.lib()
    assert L.oracle_expw(0.0) == 1.0
    assert L.oracle_expw(-np.inf) == 0.0 and L.oracle_expw(np.nan) == 0.0 and L.oracle_expw(-1e6) == 0.0
    assert L.oracle_expw(-746.0) == 0.0
    xs = rng.uniform(-745, -708.4, 20000)   # subnormal results: within one spacing of libm's
    ys = oracle.expw_array(xs)
    assert (np.abs(ys - np.exp(xs)) <= 2 * 4.94e-324 + 1e-15 * np.exp(xs)).all() and (ys > 0).all()


def test_gamma_beta_moments(oracle):
    n = 40000
    for shape in (0.5, 1.0, 2.5, 50.0, 1e5):
        g = np.array([oracle.rgamma(shape, 11, i, 1, 3) for i in range(n)])
        assert abs(g.mean() / shape - 1) < 0.03
        assert abs(g.var() / shape - 1) < 0.06
    b = np.array([oracle.rbeta(2.0, 5.0, 11, i, i, 1, 3, 4) for i in range(n)])
    assert abs(b.mean() - 2 / 7) < 0.005 and abs(b.var() - 10 / (49 * 8)) < 0.002
    assert oracle.rgamma(0.0, 1, 0, 0, 3) == 0.0


def test_update_alpha_is_positive_and_sane(oracle):
    a = np.array([oracle.update_alpha(1.0, 1.0, 1.0, 1000, 3, 99, j) for j in range(4000)])
    assert (a > 0).all()
    # E[alpha | K=3, N=1000] under Gamma(1,1) prior is well below 1 and above 0.1
    assert 0.2 < a.mean() < 0.7
