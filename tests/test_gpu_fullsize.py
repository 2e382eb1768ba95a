"""Size-independent properties of the HIP path at BASELINE.json's full sizes, where the oracle is
too slow to run: the integer sufficient statistics equal a recount from the labels, labels stay in
range, the same seed gives the same chain, different seeds do not, pi sums to one.  torch is used
only to build the synthetic matrix in HBM (as bench.py does) and to recount on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _recount(torch, X, z1, K):
    """Nk and S[k, d] recomputed from 1-based labels with plain torch reductions."""
    z = torch.as_tensor(z1.astype(np.int64) - 1, device=X.device)
    Nk = torch.bincount(z, minlength=K)[:K]
    P = X.shape[0]
    S = torch.stack([torch.bincount(z, weights=X[d].to(torch.float64), minlength=K)[:K] for d in range(P)], dim=1)
    return Nk.cpu().numpy(), S.cpu().numpy().astype(np.int64)


def _chain(bm, synth, torch, name, seed, sweeps, **kw):
    sampler, K, K_true, N, P, dseed = synth.WORKLOADS[name]
    dev = torch.device("cuda", 0)
    X, _ = synth.device_matrix(N, P, K_true, dseed, dev)
    ch = bm.Chain(sampler, N, P, K, seed=seed, **kw)
    ch.set_data_device(X.data_ptr(), keepalive=X)
    rng = np.random.default_rng(seed)
    if sampler == "collapsed":
        ch.set_initial_labels(rng.integers(1, K + 1, N).astype(np.int32))
    elif sampler == "stickbreaking":
        pi0 = np.exp(rng.random(K))
        ch.set_initial_params(pi0 / pi0.sum(), rng.random((K, P)))
    ch.sweeps(sweeps)
    return ch, X, (sampler, K, N, P)


@pytest.mark.timeout(600)
def test_c5_collapsed_K20_N1e7_P100_statistics_are_a_recount_of_the_labels():
    import torch
    import bmm_mcmc_amd as bm
    from bmm_mcmc_amd import synth
    ch, X, (_, K, N, P) = _chain(bm, synth, torch, "c5", 1000, 4)
    z = ch.labels()
    Nk, S = ch.counts()
    ch.close()
    assert z.min() >= 1 and z.max() <= K and Nk.sum() == N
    Nk2, S2 = _recount(torch, X, z, K)
    assert np.array_equal(Nk, Nk2) and np.array_equal(S, S2)
    assert (S <= Nk[:, None]).all()
    del X
    # same seed -> same chain; another seed -> another chain (first million labels compared)
    ch2, X2, _ = _chain(bm, synth, torch, "c5", 1000, 4)
    z2 = ch2.labels()
    ch2.close()
    assert np.array_equal(z, z2)
    del X2
    ch3, X3, _ = _chain(bm, synth, torch, "c5", 1001, 4)
    z3 = ch3.labels()
    ch3.close()
    assert not np.array_equal(z[:1000000], z3[:1000000])
    del X3
    # the int32 layout (X streamed as handed over) runs the same chain as the bit planes
    # (same batch: the defaulted one is rounded to each layout's own workgroup size)
    ch4, X4, _ = _chain(bm, synth, torch, "c5", 1000, 4, x_layout="int32", batch=1310720)
    ch5, X5, _ = _chain(bm, synth, torch, "c5", 1000, 4, x_layout="bits", batch=1310720)
    assert ch4.x_layout() == "int32" and ch5.x_layout() == "bits"
    z4, z5 = ch4.labels(), ch5.labels()
    ch4.close()
    ch5.close()
    assert np.array_equal(z4, z5)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("name", ["c3", "c4"])
def test_c3_dp_and_c4_stickbreaking_full_size_invariants(name):
    import torch
    import bmm_mcmc_amd as bm
    from bmm_mcmc_amd import synth
    ch, X, (sampler, K, N, P) = _chain(bm, synth, torch, name, 7, 5)
    z = ch.labels()
    Nk, S = ch.counts()
    al = ch.alpha()
    if sampler == "stickbreaking":
        pi, theta = ch.params()
        assert abs(pi.sum() - 1.0) < 1e-12 and (theta > 0).all() and (theta < 1).all()
    ch.close()
    assert z.min() >= 1 and z.max() <= K and Nk.sum() == N and al > 0
    Nk2, S2 = _recount(torch, X, z, K)
    assert np.array_equal(Nk, Nk2) and np.array_equal(S, S2)
    if sampler == "dp":
        assert (Nk > 0).sum() <= K - 1        # the truncation invariant K <= maxK - 1


@pytest.mark.timeout(600)
def test_host_matrix_uploaded_in_slabs_equals_the_device_matrix_packed_at_once():
    """bmm_chain_set_data_host has the host's cores validate and pack the int32 matrix slab by slab into pinned
    staging (here three slabs of 4 MiB of packed words, the last one ragged) and uploads only the bit planes;
    they must be the ones k_pack_bits makes from a matrix already on the device."""
    import torch
    import bmm_mcmc_amd as bm
    N, P, K = 2_500_003, 30, 6
    rng = np.random.default_rng(5)
    X = np.asfortranarray((rng.random((N, P)) < 0.35).astype(np.int32))
    z0 = rng.integers(1, K + 1, N).astype(np.int32)
    labels = []
    for how in ("host", "device"):
        with bm.Chain("collapsed", N, P, K, seed=3, batch=200_000) as c:
            if how == "host":
                c.set_data(X)
            else:
                Xd = torch.as_tensor(np.ascontiguousarray(X.T), device="cuda:0")  # [P][N] = column-major N x P
                c.set_data_device(Xd.data_ptr(), keepalive=Xd)
            c.set_initial_labels(z0)
            c.sweeps(2)
            labels.append((c.labels(), c.counts()))
    (za, (na, sa)), (zb, (nb, sb)) = labels
    assert np.array_equal(za, zb) and np.array_equal(na, nb) and np.array_equal(sa, sb)
    assert np.array_equal(sa, np.stack([X[za == k + 1].sum(axis=0) for k in range(K)]))
    bad = X.copy()
    bad[N - 2, P - 1] = 2  # a non-binary cell in the last slab is still caught
    from bmm_mcmc_amd import _capi
    with bm.Chain("collapsed", N, P, K, seed=3) as c:  # straight through the C ABI (the wrapper checks too)
        rc = _capi.lib().bmm_chain_set_data_host(c._h, _capi.vp(bad))
        assert rc == 1 and b"binary" in _capi.lib().bmm_last_error()  # BMM_E_ARG


@pytest.mark.timeout(600)
def test_posterior_recovers_the_generating_mixture_at_the_default_batch():
    """North-star acceptance on cluster proportions, at sizes the oracle cannot reach: with the default
    batch (N/4 at these sizes: bmm_default_batch) the chain must find the mixture the synthetic data were drawn
    from -- posterior-mean proportions within 0.01 of the generating weights, a draw of the labels
    agreeing with the generating ones (up to a permutation) as often as a draw from the exact posterior
    under the generating parameters would (within one point).  The DP sampler may hold a generating
    component as two clusters: a split made while the first sweep seats the observations never merges
    again under incremental Gibbs at this N (two clusters with the same theta trade members in proportion
    to their sizes -- a martingale; tools/dp_seed_scan.py: 5 of 10 seeds end with exactly the generating
    five clusters, the others with six or seven, at the default batch as at one batch per sweep).  Its
    proportions are therefore taken per generating component: every cluster counts towards the component
    its members mostly come from."""
    import torch
    import bmm_mcmc_amd as bm
    from bmm_mcmc_amd import synth
    dev = torch.device("cuda", 0)
    for sampler, K, K_true, N, P, burn, keep in (("collapsed", 3, 3, 100_000, 20, 150, 100),
                                                  ("dp", 30, 5, 400_000, 50, 150, 60)):
        X, truth = synth.device_matrix(N, P, K_true, 77, dev)
        w, theta = synth.truth(K_true, P, 77)
        ch = bm.Chain(sampler, N, P, K, seed=11)
        ch.set_data_device(X.data_ptr(), keepalive=X)
        if sampler == "collapsed":
            ch.set_initial_labels(np.random.default_rng(1).integers(1, K + 1, N).astype(np.int32))
        ch.sweeps(burn)
        counts = ch.sweeps_counts(keep)            # (keep, K) cluster sizes after each sweep, from the device
        z = ch.labels()
        ch.close()
        t = truth.cpu().numpy()
        if sampler == "collapsed":
            props = np.sort(counts / N, axis=1)[:, ::-1].mean(axis=0)
            np.testing.assert_allclose(props[:K_true], np.sort(w)[::-1], atol=0.01)
            assert props[K_true:].sum() < 0.005
        else:
            # cluster -> the generating component most of its members come from (labels of the last sweep;
            # a cluster keeps its label while it lives), then the per-sweep sizes summed per component
            owner = np.array([np.bincount(t[z == k + 1], minlength=K_true).argmax() if (z == k + 1).any() else -1
                              for k in range(K)])
            merged = np.stack([counts[:, owner == c].sum(axis=1) for c in range(K_true)], axis=1) / N
            np.testing.assert_allclose(merged.mean(axis=0), w, atol=0.01)
            big = (counts > N // 1000).sum(axis=1)   # a CRP keeps opening and closing singletons beside them
            assert (big >= K_true).all() and (big <= K_true + 2).all()
        # agreement with the generating labels up to relabelling: each found cluster -> its majority truth
        agree = sum(np.bincount(t[z == k + 1], minlength=K_true).max() for k in range(K) if (z == k + 1).any())
        # what a draw from p(z | x, generating w and theta) would score: the mean posterior mass of the truth
        th = torch.as_tensor(theta, dtype=torch.float64, device=dev)
        Xf = X.to(torch.float64)                                                  # (P, N)
        logp = torch.log(th) @ Xf + torch.log1p(-th) @ (1.0 - Xf)                 # (K_true, N)
        logp += torch.log(torch.as_tensor(w, dtype=torch.float64, device=dev))[:, None]
        post = torch.softmax(logp, dim=0)
        expected = post[truth, torch.arange(N, device=dev)].mean().item()
        assert agree / N >= expected - 0.01, (agree / N, expected)
