"""The one-chain-per-rank plumbing (bmm_mcmc_amd.multi) over gloo, world size 2, on CPU:
broadcast of the data matrix, per-rank seeds, gather of per-chain summaries.  The compute is
injected (the oracle stands in as the checker here); on a GPU node the default is the HIP chain."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_chain(sampler, X, K, nsweeps, seed, batch, **prior):
    from oracle import oracle
    Xh = np.asfortranarray(X.t().numpy())
    z0 = np.random.default_rng(seed).integers(1, K + 1, Xh.shape[0]).astype(np.int32)
    r = oracle.collapsed(Xh, z0, nsweeps + 1, K, 0.0, 0.5, 0.5, 1, 1, nsweeps, seed=seed, batch=batch or 1)
    return r["z"][0]


class _PlanesStub:
    """Stand-in for bmm_mcmc_amd.Chain in the plane broadcast (no GPU here): packs the (P, N) int32 matrix it
    is handed into ceil(P/32) words per observation, exactly the layout k_pack_bits writes."""

    def __init__(self, P, N):
        self.P, self.N = P, N
        self.W = (P + 31) // 32
        self.t = torch.zeros(self.W * N, dtype=torch.int32)
        self.saw_matrix = False
        self.filled = False

    def set_data_device(self, ptr, keepalive=None):
        X = keepalive.numpy().astype(np.uint32)                # (P, N)
        self.saw_matrix = True
        words = np.zeros((self.W, self.N), dtype=np.uint32)
        for d in range(self.P):
            words[d // 32] |= (X[d] & 1) << np.uint32(d % 32)
        self.t.copy_(torch.from_numpy(words.view(np.int32).reshape(-1)))

    def planes_tensor(self):
        return self.t

    def planes_filled(self):
        self.filled = True


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from bmm_mcmc_amd import multi
    from util import synth
    w, r, l = multi.init(backend="gloo")
    assert (w, r) == (world, rank)
    P, N, K = 8, 600, 3
    if rank == 0:
        Xh, _, _, _ = synth(N, P, K, 18)
        X = torch.from_numpy(np.ascontiguousarray(Xh.T))  # (P, N) = N x P column-major
    else:
        X = torch.zeros((P, N), dtype=torch.int32)
    multi.broadcast_data(X, src=0)
    # the collective bench.py and the multi-GPU launcher use: rank 0 packs, the planes travel, nobody else
    # ever holds the matrix
    stub = _PlanesStub(P, N)
    multi.broadcast_planes(stub, X if rank == 0 else None, src=0)
    planes = (int(stub.t.to(torch.int64).sum()), stub.saw_matrix, stub.filled)
    z, summ = multi.run_chains("collapsed", X, K, 30, base_seed=1000, batch=64, run_fn=_oracle_chain)
    z_same, _ = multi.run_chains("collapsed", X, K, 30, base_seed=1000, batch=64, run_fn=_oracle_chain)
    mx = multi.max_over_ranks(1.0 + rank)
    q.put((rank, int(X.sum()), z.tolist(), z_same.tolist(), summ.tolist(), mx, multi.chain_seed(1000, rank), planes))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_broadcast_seed_and_gather(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, sum0, z0, z0b, summ0, mx0, seed0, pl0), (r1, sum1, z1, z1b, summ1, mx1, seed1, pl1) = got
    assert pl0[0] == pl1[0] != 0                # both ranks hold the same bit planes
    assert pl0[1:] == (True, False)             # rank 0 packed the matrix it generated
    assert pl1[1:] == (False, True)             # rank 1 received planes only and declared them filled
    assert sum0 == sum1 and sum0 > 0            # every rank holds the broadcast matrix
    assert (seed0, seed1) == (1000, 1001)       # chain c runs under key base + c
    assert z0 == z0b and z1 == z1b              # same seed -> same chain
    assert z0 != z1                             # different seeds -> different chains
    assert summ0 == summ1 and len(summ0) == 2   # all ranks see every chain's summary, in rank order
    np.testing.assert_allclose(np.sum(summ0, axis=1), 1.0)
    assert all(a >= b for a, b in zip(summ0[0], summ0[0][1:]))  # sorted proportions
    assert mx0 == mx1 == 2.0


def _worker4(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from bmm_mcmc_amd import multi
    from util import synth
    multi.init(backend="gloo")
    P, N, K = 6, 300, 3
    Xh, _, _, _ = synth(N, P, K, 21)
    X = torch.from_numpy(np.ascontiguousarray(Xh.T)) if rank == 0 else torch.zeros((P, N), dtype=torch.int32)
    stub = _PlanesStub(P, N)
    multi.broadcast_planes(stub, X if rank == 0 else None, src=0)        # planes from rank 0 to three receivers
    multi.broadcast_data(X, src=0)
    zs, summ = multi.run_chains("collapsed", X, K, 12, base_seed=500, batch=32, run_fn=_oracle_chain, chains_per_rank=2)
    q.put((rank, [z.tolist() for z in zs], summ.tolist(), int(stub.t.to(torch.int64).sum()), stub.filled,
           multi.max_over_ranks(10.0 - rank)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_four_ranks_two_chains_each(oracle):
    """world size 4, two chains per rank: chain r * 2 + c of the job runs under key base + r * 2 + c (the
    numbering of bmm_multi_run), every rank receives the planes and sees all eight summaries in job order"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from util import synth
    Xh, _, _, _ = synth(300, 6, 3, 21)
    Xt = torch.from_numpy(np.ascontiguousarray(Xh.T))
    summaries = [g[2] for g in got]
    assert all(s == summaries[0] for s in summaries) and len(summaries[0]) == 8
    planes = {g[3] for g in got}
    assert len(planes) == 1 and 0 not in planes                     # the same planes on all four ranks
    assert [g[4] for g in got] == [False, True, True, True]         # receivers declared theirs filled
    assert all(g[5] == 10.0 for g in got)
    seen = []
    for rank, zs, _, _, _, _ in got:
        assert len(zs) == 2
        for c, z in enumerate(zs):
            want = _oracle_chain("collapsed", Xt, 3, 12, 500 + rank * 2 + c, 32)     # key = base + job index
            assert z == want.tolist()
            props = np.sort(np.bincount(np.array(z) - 1, minlength=3) / 300)[::-1]
            np.testing.assert_allclose(summaries[0][rank * 2 + c], props)            # row = job index
            seen.append(tuple(z))
    assert len(set(seen)) == 8                                       # eight different chains


def test_bench_refuses_more_gpus_than_there_are():
    """`python bench.py --gpus N` started without torch.distributed.run drives N devices from one process; with
    fewer devices visible it must exit non-zero and say why (a SCALE run on a short node must not report numbers)"""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--no-cpu", "--no-extra"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "64 GPUs asked for" in r.stderr and "visible" in r.stderr
    assert r.stdout.strip() == ""                                    # no JSON line


def test_single_process_helpers_need_no_group():
    from bmm_mcmc_amd import multi
    assert multi.world() == (1, 0, 0) or multi.world()[0] >= 1
    assert multi.max_over_ranks(3.5) == 3.5
    assert multi.gather_summaries([0.25, 0.75]).tolist() == [[0.25, 0.75]]
    assert multi.chain_seed(2 ** 64 - 1, 2) == 1
