"""Several chains in one call and on one device (include/bmm_mcmc.h: bmm_multi_run, bmm_chain_share_data,
bmm_chains_sweeps) -- through the C ABI on one GPU: chains on devices {0, 0, ...} need no collective, share
one copy of the bit planes and run on their own streams from their own host threads; each must still be its
oracle chain (seed + c) bit for bit.  The RCCL broadcast itself is exercised on the one device the box has by
bmm_multi_selfcheck (library opened, communicator built, ncclBroadcast run, words compared)."""
import ctypes as C

import numpy as np
import pytest

import bmm_mcmc_amd as bm
from bmm_mcmc_amd import _capi
from util import synth

pytestmark = pytest.mark.gpu


def _z0(N, K, seed):
    return np.random.default_rng(seed).integers(1, K + 1, N).astype(np.int32)


def test_two_collapsed_chains_on_one_device_equal_their_oracle_chains(oracle):
    X, _, _, _ = synth(6000, 40, 5, 12)
    N, K, seed = 6000, 8, 500
    z0s = [_z0(N, K, 1), _z0(N, K, 2)]
    got = bm.gibbs_collapsed(X, 9, K, burnin=2, seed=seed, batch=1000, chains=2, devices=[0, 0], initial_K=z0s)
    assert isinstance(got, list) and len(got) == 2
    for c in range(2):
        want = oracle.collapsed(X, z0s[c], 9, K, 0.0, 0.5, 0.5, 1, 1, 2, seed=seed + c, batch=1000)
        for k in ("z", "theta", "alpha"):
            assert np.array_equal(got[c][k], want[k], equal_nan=True), (c, k)
        assert got[c]["permutations"].shape == (7, K)
    assert not np.array_equal(got[0]["z"], got[1]["z"])   # different keys, different chains


def test_four_chains_of_the_other_samplers(oracle):
    X, _, _, _ = synth(4000, 30, 4, 7)
    got = bm.gibbs_dp(X, 7, burnin=1, maxK=12, seed=40, batch=250, chains=4)          # devices=None: all on 0
    for c in range(4):
        want = oracle.dp(X, 7, 0.0, 0.5, 0.5, 1, 1, 1, 12, seed=40 + c, batch=250)
        for k in ("z", "theta", "alpha"):
            assert np.array_equal(got[c][k], want[k], equal_nan=True), (c, k)
    rng = np.random.default_rng(3)
    pis = [rng.dirichlet(np.ones(9)) for _ in range(3)]
    ths = [rng.random((9, 30)) for _ in range(3)]
    got = bm.gibbs_stickbreaking(X, 6, 9, burnin=0, seed=11, chains=3, devices=[0, 0, 0], initial_pi=pis,
                                 initial_theta=ths)
    for c in range(3):
        want = oracle.stickbreaking(X, pis[c], ths[c], 6, 9, 0.0, 0.5, 0.5, 1, 1, 0, seed=11 + c)
        for k in ("pi", "z", "theta", "alpha"):
            assert np.array_equal(got[c][k], want[k], equal_nan=True), (c, k)


def test_multi_run_argument_errors():
    X, _, _, _ = synth(300, 6, 2, 1)
    with pytest.raises(ValueError, match="one device per chain"):
        bm.gibbs_dp(X, 5, seed=1, chains=2, devices=[0])
    with pytest.raises(bm.BmmError, match="out of range"):
        bm.gibbs_dp(X, 5, seed=1, chains=2, devices=[0, 99])


def test_native_plane_broadcast_argument_checks():
    """bmm_chains_broadcast_planes on what one GPU allows: one chain (nothing to send), and the refusal of
    two chains on one device (those share, they do not broadcast)."""
    X, _, _, _ = synth(2000, 10, 2, 1)
    with bm.Chain("collapsed", 2000, 10, 3, seed=1) as a, bm.Chain("collapsed", 2000, 10, 3, seed=2) as b:
        with pytest.raises(bm.BmmError, match="holds no bit planes"):
            bm.broadcast_planes([a])
        a.set_data(X)
        bm.broadcast_planes([a])
        with pytest.raises(bm.BmmError, match="one chain per device"):
            bm.broadcast_planes([a, b])


def test_rccl_broadcast_selfcheck_on_this_device():
    dev = (C.c_int * 1)(0)
    _capi.check(_capi.lib().bmm_multi_selfcheck(C.c_int(1), dev, C.c_int64(1 << 20)))


def test_resident_chains_sharing_planes_and_swept_together(oracle):
    """chains_per_gpu: four resident chains over ONE copy of the bit planes, advanced by one host thread each."""
    N, P, K = 20000, 50, 20
    X, _, _, _ = synth(N, P, 6, 5)
    chains = [bm.Chain("collapsed", N, P, K, batch=2500, seed=900 + c) for c in range(4)]
    try:
        chains[0].set_data(X)
        for c in chains[1:]:
            c.share_data(chains[0])
        z0s = [_z0(N, K, 30 + c) for c in range(4)]
        for c, z0 in zip(chains, z0s):
            c.set_initial_labels(z0)
        bm.sweep_chains(chains, 3)
        bm.sweep_chains(chains, 2)
        for c in chains:
            c.sync()
        for i, c in enumerate(chains):
            want = oracle.collapsed(X, z0s[i], 6, K, 0.0, 0.5, 0.5, 1, 1, 5, seed=900 + i, batch=2500)
            assert np.array_equal(c.labels(), want["z"][0]), i
            assert c.alpha() == want["alpha"][0, 0]
        with pytest.raises(bm.BmmError, match="already has its data"):
            chains[1].share_data(chains[0])
        # the planes are reference-counted: the chain that packed them may go first
        chains[0].close()
        bm.sweep_chains(chains[1:], 2)
        for c in chains[1:]:
            c.sync()
        want = oracle.collapsed(X, z0s[3], 8, K, 0.0, 0.5, 0.5, 1, 1, 7, seed=903, batch=2500)
        assert np.array_equal(chains[3].labels(), want["z"][0])
    finally:
        for c in chains:
            c.close()


def test_planes_handed_over_by_the_caller_equal_a_packed_matrix(oracle):
    """What a rank of the multi-process launcher does after the broadcast: it never sees the int32 matrix."""
    import torch
    from bmm_mcmc_amd import multi
    N, P, K = 5000, 70, 6
    X, _, _, _ = synth(N, P, 4, 3)
    z0 = _z0(N, K, 8)
    with bm.Chain("collapsed", N, P, K, batch=700, seed=5) as src, bm.Chain("collapsed", N, P, K, batch=700, seed=5) as dst:
        src.set_data(X)
        a, n = src.planes()
        b, n2 = dst.planes()
        assert n == n2 == 3 * N
        multi.device_ints(b, n, torch.device("cuda", 0)).copy_(multi.device_ints(a, n, torch.device("cuda", 0)))
        dst.planes_filled()            # waits for the device: the copy above ran on torch's stream
        for c in (src, dst):
            c.set_initial_labels(z0)
            c.sweeps(4)
        assert np.array_equal(src.labels(), dst.labels())
        want = oracle.collapsed(X, z0, 5, K, 0.0, 0.5, 0.5, 1, 1, 4, seed=5, batch=700)
        assert np.array_equal(dst.labels(), want["z"][0])


def test_set_data_device_waits_for_the_producer(oracle):
    """bmm_chain_set_data_device right behind kernels still writing X on another stream: the
    chain must pack the finished matrix, not what was there before."""
    import torch
    N, P, K = 400000, 24, 4
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Xt = torch.zeros((P, N), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(20):                                   # a queue of work ahead of the final values
            Xt.copy_((torch.rand((P, N), generator=g, device=dev) < 0.3).to(torch.int32))
        with bm.Chain("collapsed", N, P, K, seed=1) as ch:
            ch.set_data_device(Xt.data_ptr(), keepalive=Xt)   # no synchronize in between
            z0 = _z0(N, K, 2)
            ch.set_initial_labels(z0)
            ch.sweeps(0)
            nk, s = ch.counts()
    Xh = Xt.cpu().numpy().T
    assert np.array_equal(nk, np.bincount(z0 - 1, minlength=K))
    assert np.array_equal(s, np.stack([Xh[z0 == k + 1].sum(axis=0) for k in range(K)]))


@pytest.mark.timeout(600)
def test_bench_line_of_a_two_rank_job_rehearsed_on_one_gpu():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), rehearsed with two
    ranks on the one device there is (BMM_BENCH_REHEARSE=1: gloo instead of RCCL -- the code path, never a
    measurement): plane broadcast, per-rank seeds, collectives called by every rank, max over ranks, ONE JSON
    line from rank 0 with the multi_gpu block a scaling record can be checked against."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BMM_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6",
                        "--warmup", "2", "--burn", "4", "--workload", "c2", "--no-cpu"],
                       capture_output=True, text=True, timeout=500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                           # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6 and d["warmup"] == 2
    m = d["multi_gpu"]
    assert m["ranks_seen_by_the_collective"] == 2 and m["world_size"] == 2 and len(m["per_rank_sweeps_per_s"]) == 2
    assert "REHEARSAL" in m["launcher"] and m["plane_bytes"] == 4 * 100_000
    assert d["config"]["chains"] == 2 and d["value"] > 0
    assert abs(d["value"] - 2 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]    # value = chains x steps / max time


def test_multi_device_bookkeeping_on_fake_devices(oracle, dbg_lib):
    """The test variant's BMM_DEBUG_FAKE_DEVICES=2: device indices 0 and 1 are both the one GPU, kept apart wherever
    the library reasons about devices, the RCCL broadcast between them replaced by a copy.  bmm_multi_run with
    devices = {0, 1, 0, 1} then runs everything it would run on two GPUs except the collective itself: placement,
    a holder of the planes per device, receivers declaring theirs filled, chains 2 and 3 sharing with the holder of
    THEIR device, one host thread per chain -- and every chain is its oracle chain (seed + c)."""
    dbg_lib.setenv("BMM_DEBUG_FAKE_DEVICES", "2")
    X, _, _, _ = synth(5000, 24, 3, 6)
    z0s = [_z0(5000, 3, 10 + c) for c in range(4)]
    outs = bm.gibbs_collapsed(X, 8, 3, burnin=2, seed=40, batch=700, chains=4, devices=[0, 1, 0, 1], initial_K=z0s)
    for c, out in enumerate(outs):
        want = oracle.collapsed(X, z0s[c], 8, 3, 0.0, 0.5, 0.5, 1, 1, 2, seed=40 + c, batch=700)
        assert np.array_equal(out["z"], want["z"]) and np.array_equal(out["theta"], want["theta"], equal_nan=True), c
    outs = bm.gibbs_dp(X, 7, burnin=1, maxK=9, seed=3, batch=300, chains=3, devices=[1, 0, 1])   # device 1 named first: it packs
    for c, out in enumerate(outs):
        want = oracle.dp(X, 7, 0.0, 0.5, 0.5, 1, 1, 1, 9, seed=3 + c, batch=300)
        assert np.array_equal(out["z"], want["z"]), c
    with pytest.raises(bm.BmmError, match="out of range"):
        bm.gibbs_collapsed(X, 4, 3, seed=1, chains=2, devices=[0, 2])
    # resident chains, one per "device": the second receives the first one's planes (bmm_chains_broadcast_planes)
    a, b = bm.Chain("collapsed", 5000, 24, 3, seed=7, batch=512, device=0), bm.Chain("collapsed", 5000, 24, 3, seed=8, batch=512, device=1)
    third = bm.Chain("collapsed", 5000, 24, 3, seed=9, batch=512, device=1)
    try:
        a.set_data(X)
        bm.broadcast_planes([a, b])
        with pytest.raises(bm.BmmError, match="different devices"):
            third.share_data(a)                      # a lives on device 0, third on device 1
        third.share_data(b)
        for ch, s in ((a, 7), (b, 8), (third, 9)):
            ch.set_initial_labels(z0s[0])
        bm.sweep_chains([a, b, third], 4)
        for ch, s in ((a, 7), (b, 8), (third, 9)):
            want = oracle.collapsed(X, z0s[0], 5, 3, 0.0, 0.5, 0.5, 1, 1, 4, seed=s, batch=512)
            assert np.array_equal(ch.labels(), want["z"][0]), s
        with pytest.raises(bm.BmmError, match="one chain per device"):
            bm.broadcast_planes([a, b, third])
    finally:
        for ch in (third, b, a):
            ch.close()


@pytest.mark.timeout(600)
def test_bench_line_contract_on_one_gpu():
    """the ONE JSON line `python bench.py` prints: the keys the driver reads, the roofline and cpu_baseline objects"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c2", "--steps", "8", "--warmup", "2", "--burn", "4",
                        "--no-extra", "--cpu-seconds", "1", "--cpu-rows", "20000"], capture_output=True, text=True, timeout=500,
                       env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert (d["n_gpus"], d["steps"], d["warmup"], d["higher_is_better"], d["scaling"], d["vs_baseline"]) == (1, 8, 2, True, "weak", None)
    assert d["unit"] == "sweeps/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 8e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["kernel_ms_per_sweep"] <= d["ms_per_step"]               # the kernel's time is part of the sweep's
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
