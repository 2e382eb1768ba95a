"""One independent MCMC chain per GPU of a node.

The path shards across chains, not within one: every rank holds a full replica of the data
matrix and runs its own chain, so there is exactly one collective on the data path -- the
one-off broadcast of X from rank 0 (RCCL over xGMI with the "nccl" backend) -- plus an
optional gather of small per-chain summaries at the end.  Launch one process per GPU
(`python -m torch.distributed.run --nproc-per-node N ...`).

torch is plumbing here (process group, device tensors).  The chain itself is the HIP path
(bmm_mcmc_amd.Chain); `run_fn` lets the multi-process CPU tests drive the same plumbing
over gloo with a checker of their own.
"""
import os

import numpy as np

LOCAL_DEVICE = None  # set by a caller that does not map LOCAL_RANK to the device index (bench.py's rehearsal mode)


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(
        os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, device=None):
    """Join the process group if launched with more than one rank.  Returns (world, rank, local)."""
    import torch.distributed as dist
    w, r, l = world()
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # under torch.distributed.run
    if (w > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return w, r, l


def chain_seed(base, rank):
    """Chain c of a job gets Philox key base + c (chains are independent streams)."""
    return (int(base) + int(rank)) & 0xFFFFFFFFFFFFFFFF


def broadcast_data(X, src=0):
    """Broadcast the (P, N) int32 tensor X (= N x P column-major) from `src` to every rank, in place, and
    wait for it: whoever reads X next may do so on any stream.  (4 GB at K=20, N=1e7, P=100: prefer
    broadcast_planes, which moves what a rank actually keeps.)"""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(X, src=src)
        if X.is_cuda:
            torch.cuda.synchronize(X.device)
    return X


def broadcast_planes(chain, X=None, src=0):
    """The one collective of the chain path, on what a rank keeps: rank `src` hands its int32 matrix X
    (a (P, N) device tensor) to its chain, which packs it into bit planes; the planes -- 4 * ceil(P/32)
    bytes per observation (160 MB instead of 4 GB at K=20, N=1e7, P=100) -- are broadcast into the other
    ranks' chains, which never see the matrix.  Single rank: just the hand-over."""
    import torch
    import torch.distributed as dist
    w, r, l = world()
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not multi or r == src:
        if X is None:
            raise ValueError("the source rank needs the matrix")
        chain.set_data_device(X.data_ptr(), keepalive=X if hasattr(chain, "planes_tensor") else None)  # packed now
    if not multi:
        return
    if hasattr(chain, "planes_tensor"):       # a stand-in chain of the CPU tests hands over its own tensor
        t, dev = chain.planes_tensor(), None
    else:
        ptr, n = chain.planes()
        dev = torch.device("cuda", l if LOCAL_DEVICE is None else LOCAL_DEVICE)
        t = device_ints(ptr, n, dev)
    dist.broadcast(t, src=src)
    if dev is not None:
        torch.cuda.synchronize(dev)
    if r != src:
        chain.planes_filled()


def gather_summaries(vec):
    """All-gather one small float64 vector per chain; returns a (world, len) array on every rank."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.ascontiguousarray(vec, dtype=np.float64))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t.numpy()[None, :]
    if dist.get_backend() == "nccl":
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def max_over_ranks(x):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def run_chains(sampler, X, K, nsweeps, base_seed=1000, batch=None, run_fn=None, chains_per_rank=1, **prior):
    """Run `chains_per_rank` chains on this rank over the (already broadcast) data and gather the sorted cluster
    proportions of every chain of the job.  Chain c of rank r is chain r * chains_per_rank + c of the job and
    runs under key base_seed + that index (bmm_multi_run's numbering: seed + c).
    `run_fn(sampler, X, K, nsweeps, seed, batch, **prior)` must return final 1-based labels; the default is
    the HIP chain on this rank's GPU.  Returns (labels -- an array, or a list of arrays when chains_per_rank > 1 --
    and the (world * chains_per_rank, K) summaries, rows in job order, identical on every rank)."""
    w, r, l = world()
    if run_fn is None:
        run_fn = _hip_chain
    zs, props = [], []
    for c in range(int(chains_per_rank)):
        seed = chain_seed(base_seed, r * int(chains_per_rank) + c)
        z = np.asarray(run_fn(sampler, X, K, nsweeps, seed, batch, **prior))
        zs.append(z)
        props.append(np.sort(np.bincount(z[z > 0] - 1, minlength=K)[:K] / max(1, (z > 0).sum()))[::-1])
    summ = gather_summaries(np.concatenate(props)).reshape(-1, K)  # rank-major, then chain: job order
    return (zs[0] if chains_per_rank == 1 else zs), summ


def _hip_chain(sampler, X, K, nsweeps, seed, batch, **prior):
    from . import Chain
    w, r, l = world()
    P, N = X.shape
    with Chain(sampler, N, P, K, batch=batch, seed=seed, device=l, **prior) as ch:
        ch.set_data_device(X.data_ptr(), keepalive=X)
        rng = np.random.default_rng(seed)
        if sampler == "collapsed":
            ch.set_initial_labels(rng.integers(1, K + 1, N).astype(np.int32))
        elif sampler == "stickbreaking":
            pi0 = np.exp(rng.random(K))
            ch.set_initial_params(pi0 / pi0.sum(), rng.random((K, P)))
        ch.sweeps(nsweeps)
        return ch.labels()


# ---------------------------------------------------------------- one chain over several ranks
class _DeviceInts:
    """int32 device memory owned by the library, exposed through the CUDA array interface."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False),
                                         "version": 2}


def device_ints(ptr, n, device):
    """Zero-copy torch view of `n` int32 at device address `ptr`."""
    import torch
    return torch.as_tensor(_DeviceInts(ptr, n), device=device)


class ShardedChain:
    """One stick-breaking / full chain whose observations are split over the ranks of the job
    (SURVEY.md section 8 row f4).  Rank r holds rows [first_row, first_row + n_local) of the
    N_total; every rank passes the SAME seed and initial (pi, theta).  Per sweep: local z-resample,
    one all-reduce of the K*(P+1) integer statistic deltas (RCCL with the nccl backend), then the
    parameter draws, which come out identical on every rank."""

    def __init__(self, sampler, X_local, N_total, first_row, K, pi0, theta0, seed, device=0, **prior):
        import torch
        from . import Chain
        P, n_local = X_local.shape
        self.K, self.P = int(K), int(P)
        self.dev = torch.device("cuda", device)
        self.chain = Chain(sampler, n_local, P, K, seed=seed, device=device, **prior)
        self.chain.set_data_device(X_local.data_ptr(), keepalive=X_local)
        self.chain.set_shard(N_total, first_row)
        self.chain.set_initial_params(pi0, theta0)
        a, b = self.chain.shard_deltas()
        self.d_nk = device_ints(a, K, self.dev)
        self.d_s = device_ints(b, K * P, self.dev)
        # the chain's own HIP stream as a torch stream: collectives issued under it are ordered behind
        # the resample kernel and ahead of the parameter draws by the streams alone
        self.ext = torch.cuda.ExternalStream(self.chain.stream(), device=self.dev)

    def sweep(self):
        """One sweep with no host round trip: resample (enqueued), the two all-reduces and the parameter
        draws all ordered on the chain's stream."""
        import torch
        import torch.distributed as dist
        self.chain.shard_resample(wait=False)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            with torch.cuda.stream(self.ext):
                dist.all_reduce(self.d_nk)
                dist.all_reduce(self.d_s)
        self.chain.shard_finish()

    def close(self):
        self.chain.close()
