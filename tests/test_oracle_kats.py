"""RNG-free known answers (tests/golden/kats.json, NumPy from the reference formulas)
against both restatements in the oracle.  PARITY UNPINNED: the reference ships no
golden vectors; these pin the arithmetic to its cited source lines only."""
import numpy as np
import pytest

from util import load_dataset, load_kats

RTOL = 1e-12


@pytest.mark.parametrize("name", ["K2_N100_P5", "K2_N1000_P5", "K3_N1000_P5"])
@pytest.mark.parametrize("spec", [False, True], ids=["literal", "spec"])
def test_conditionals(oracle, name, spec):
    kats = load_kats()
    rec = kats["datasets"][name]
    X = load_dataset(name)
    N, P, K = rec["N"], rec["P"], rec["K"]
    assert X.shape == (N, P)
    z = (1 + (np.arange(N) % K)).astype(np.int32)
    al, be, ga = kats["alpha"], kats["beta"], kats["gamma"]
    for case in rec["cases"]:
        i = case["i"]
        assert X[i].tolist() == case["x"]
        raw, norm = oracle.collapsed_cond(X, z, i, K, al, be, ga, spec=spec)
        np.testing.assert_allclose(norm, case["collapsed_norm"], rtol=RTOL)
        if not spec:
            np.testing.assert_allclose(raw, case["collapsed_raw"], rtol=RTOL)
        else:  # spec returns log scores
            np.testing.assert_allclose(np.exp(raw), case["collapsed_raw"], rtol=1e-11)
        logw, dnorm = oracle.dp_cond(X, z, i, K, al, be, ga, spec=spec)
        np.testing.assert_allclose(logw, case["dp_logw"], rtol=RTOL)
        np.testing.assert_allclose(dnorm, case["dp_norm"], rtol=1e-11)
        sraw, snorm = oracle.sb_cond(X, i, np.array(rec["pi_true"]), np.array(rec["theta_true"]), spec=spec)
        np.testing.assert_allclose(snorm, case["sb_norm"], rtol=1e-11)
        if not spec:
            np.testing.assert_allclose(sraw, case["sb_raw"], rtol=RTOL)


def test_fixture_checksums():
    # element sums and block structure of the bundled data
    X = load_dataset("K2_N100_P5")
    assert X.sum() == 212 and X[99].tolist() == [0, 0, 1, 1, 0]
    np.testing.assert_allclose(X[:70].mean(axis=0), [0.686, 0.786, 0.129, 0.129, 0.129], atol=1e-3)
    assert load_dataset("K2_N1000_P5").sum() == 2128
    assert load_dataset("K3_N1000_P5").sum() == 2172


def test_empty_cluster_has_probability_zero(oracle):
    # collapsed_gibbs.cpp:104,131-133: a cluster with no other member gets exactly 0
    X = load_dataset("K2_N100_P5")
    z = np.ones(100, dtype=np.int32)
    z[0] = 2  # observation 0 alone in cluster 2
    for spec in (False, True):
        _, norm = oracle.collapsed_cond(X, z, 0, 3, 1.0, 0.5, 0.5, spec=spec)
        assert norm[1] == 0.0 and norm[2] == 0.0 and norm[0] == 1.0
