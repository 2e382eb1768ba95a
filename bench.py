#!/usr/bin/env python3
"""Gibbs sweeps/s of the cluster-allocation path on N MI355X GPUs (one chain per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one Gibbs sweep: every z_n resampled once, sufficient statistics and
parameters refreshed.  Default workload "c5" is BASELINE.json's HBM-roofline
configuration, one chain per GPU: gibbs_collapsed, K=20, N=1e7, P=100, synthetic.
The data matrix is generated in HBM on rank 0 and broadcast over RCCL (the only
collective on this path); chains are independent, so scaling is weak.

Prints ONE JSON line on rank 0.  `roofline` is for the z-resample kernel, timed with HIP
events on the chain's own stream inside the timed region.  Algorithmic bytes follow SURVEY.md
section 8d for the layout ACTUALLY streamed: by default X is packed once into bit planes when
it is handed over, so a sweep streams N*(4*ceil(P/32)+8) bytes (bit-plane words + z read + z
write); the int32-equivalent figure N*(4P+8) is reported only as a labelled secondary, and the
kernel that streams the int32 matrix as R hands it over (--x-layout int32) is measured beside
the headline in `other_workloads`.
`cpu_baseline` is the oracle's sufficient-statistics chain (same batch semantics) on the
host cores of this box, on a bounded row sample, scaled to sweeps/s at the workload's N.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--burn", type=int, default=30,
                    help="untimed set-up sweeps before the warm-up, so that the timed region is the "
                         "chain's steady state and not its first sweeps from a random allocation")
    ap.add_argument("--workload", default="c5", choices=["c2", "c3", "c4", "c5", "ns"])
    ap.add_argument("--batch", type=int, default=0, help="observations per frozen-statistics batch (0 = default)")
    ap.add_argument("--n", type=int, default=0, help="override N (debug)")
    ap.add_argument("--k", type=int, default=0, help="override K (debug)")
    ap.add_argument("--x-layout", default="bits", choices=["bits", "int32"],
                    help="what the resample kernel streams: bit planes packed once at hand-over (default) "
                         "or the int32 matrix as R hands it over")
    ap.add_argument("--event-stride", type=int, default=4, help="time the resample launches of every n-th sweep")
    ap.add_argument("--no-events", action="store_true",
                    help="debug: no HIP events around the resample launches (roofline is then null)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the c2 / north-star side measurements")
    ap.add_argument("--shard", action="store_true",
                    help="ONE chain whose rows are split over the ranks (stick-breaking / full workloads "
                         "only; strong scaling, one all-reduce of the statistics per sweep)")
    ap.add_argument("--cpu-rows", type=int, default=1_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline leg")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import bmm_mcmc_amd as bm
    from bmm_mcmc_amd import multi, synth

    world, rank, local = multi.world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi.init("nccl", device=dev)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def measure(workload, steps, warmup, burn, batch_arg, n_override=0, k_override=0, x_layout="bits"):
        """One chain per rank on `workload`; returns timing of `steps` sweeps (max over ranks)."""
        sampler, K, K_true, N, P, dseed = synth.WORKLOADS[workload]
        if n_override:
            N = n_override
        if k_override:
            K = k_override
            K_true = min(K_true, K)
        # data: generated in HBM on rank 0, broadcast once over RCCL/xGMI
        if rank == 0:
            X, _ = synth.device_matrix(N, P, K_true, dseed, dev)
        else:
            X = torch.empty((P, N), dtype=torch.int32, device=dev)
        multi.broadcast_data(X, src=0)
        torch.cuda.synchronize()
        seed = multi.chain_seed(1000, rank)  # chain seeds 1000 + c
        ch = bm.Chain(sampler, N, P, K, alpha=None, beta=0.5, gamma=0.5, a=1, b=1,
                      batch=batch_arg if batch_arg > 0 else None, seed=seed, device=local, x_layout=x_layout)
        ch.set_data_device(X.data_ptr(), keepalive=X)
        rng = np.random.default_rng(seed)
        if sampler == "collapsed":
            ch.set_initial_labels(rng.integers(1, K + 1, N).astype(np.int32))
        elif sampler in ("stickbreaking", "full"):
            pi0 = np.exp(rng.random(K))
            ch.set_initial_params(pi0 / pi0.sum(), rng.random((K, P)))
        ch.sweeps(burn)
        ch.sweeps(warmup)
        ch.sync()
        # HIP events around the resample launches of every 4th timed sweep: the per-launch average
        # is what the roofline needs, and 16 event records per sweep would cost ~7 % of the sweep
        ch.profile(0 if args.no_events else args.event_stride)
        barrier()
        t0 = time.perf_counter()
        ch.sweeps(steps)
        ch.sync()
        barrier()
        t1 = time.perf_counter()
        kern_ms, kern_n = ch.profile_read()
        ch.profile(False)
        # launches per sweep are structural: one per batch (one in all for the explicit-parameter samplers)
        lps = 1 if sampler in ("stickbreaking", "full") else -(-N // ch.batch)
        per_launch = (kern_ms / kern_n) if kern_n else 0.0
        kern_ms = multi.max_over_ranks(per_launch * lps * steps)  # resample-kernel ms over the timed sweeps
        kern_n = lps * steps
        m = {"sampler": sampler, "K": K, "N": N, "P": P, "batch": ch.batch, "shape": ch.kernel_shape(),
             "dt": multi.max_over_ranks(t1 - t0), "kern_ms": kern_ms, "kern_n": kern_n,
             "X": X, "chain": ch, "layout": ch.x_layout()}
        return m

    def copy_rate():
        """GB/s (read + write) of a plain device-to-device copy in this run, on this box: the achievable
        HBM rate SURVEY.md 8d asks to report beside the 8 TB/s spec."""
        n = 1 << 28  # 1 GiB of int32 each way
        a = torch.empty(n, dtype=torch.int32, device=dev)
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        del a, b
        torch.cuda.empty_cache()
        return 2 * n * 4 / (ms * 1e-3) / 1e9

    def sweep_bytes(N, P, layout):
        """bytes of the layout streamed + 4 B old label + 4 B new label, per observation."""
        return N * ((4 * ((P + 31) // 32) if layout == "bits" else 4 * P) + 8)

    if args.shard:
        return bench_sharded(args, world, rank, local, dev, barrier)
    m = measure(args.workload, args.steps, args.warmup, args.burn, args.batch, args.n, args.k, args.x_layout)
    sampler, K, N, P, batch, shape = m["sampler"], m["K"], m["N"], m["P"], m["batch"], m["shape"]
    dt, kern_ms, kern_n, X, ch = m["dt"], m["kern_ms"], m["kern_n"], m["X"], m["chain"]

    result = None
    if rank == 0:
        layout = m["layout"]
        bytes_per_sweep = sweep_bytes(N, P, layout)
        achieved = (bytes_per_sweep * args.steps) / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
        eq_bytes = sweep_bytes(N, P, "int32")
        eq_achieved = (eq_bytes * args.steps) / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload + ("" if layout == "int32" else "_bits"))
            except Exception:
                traffic = None
        result = {
            "metric": "gibbs_sweeps_per_s",
            "value": world * args.steps / dt,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: gibbs_%s K=%d N=%d P=%d, 1 chain per GPU" % (args.workload, sampler, K, N, P),
                       "sampler": sampler, "K": K, "N": N, "P": P, "batch": batch, "chains": world,
                       "burn_sweeps": args.burn,
                       "x_layout": ("bit planes, ceil(P/32) words per observation, packed once at hand-over"
                                    if layout == "bits" else "int32 column-major (as R hands it over)"),
                       "allocations_per_s": world * args.steps * N / dt},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": "k_resample", "kernel_ms_per_sweep": kern_ms / args.steps,
                         "launches_per_sweep": kern_n / args.steps,
                         "algorithmic_bytes_per_sweep": bytes_per_sweep,
                         "lds_bytes": shape["lds_bytes"], "threads": shape["threads"]},
        }
        copy_gbps = copy_rate() if world == 1 else None
        if copy_gbps:
            result["roofline"]["copy_kernel_GBps"] = copy_gbps
            result["roofline"]["frac_of_copy_kernel"] = achieved / copy_gbps if achieved else None
        if layout == "bits":
            # labelled secondary: what an int32-streaming kernel would have to move in this time
            result["roofline"]["layout"] = "bit planes"
            result["roofline"]["int32_equivalent"] = {
                "bytes_per_sweep": eq_bytes, "GBps": eq_achieved,
                "frac_of_hbm_peak": (eq_achieved / HBM_PEAK_GBS) if eq_achieved else None,
                "note": "secondary figure, not a roofline fraction: the packed kernel does not move these bytes"}
            result["roofline"]["note"] = ("with X in bit planes the kernel streams 17x fewer bytes than the int32 "
                                          "layout and is bound by fp64 VALU issue and LDS lookups, not by HBM; the "
                                          "HBM-bound kernel on the layout R hands over is other_workloads.c5_int32")
        if not args.no_cpu and world == 1:
            from oracle import oracle
            rows = min(args.cpu_rows, N)
            Xh = np.asfortranarray(X[:, :rows].t().cpu().numpy())  # first `rows` shuffled rows
            threads = min(os.cpu_count() or 1, 16)
            cb = max(1, min(batch, rows))
            probe = oracle.time_sweeps(sampler, Xh, K, 1, cb, 1000, threads)  # size the leg to ~cpu-seconds
            cpu_sweeps = int(max(2, min(200, round(args.cpu_seconds / max(probe, 1e-3)))))
            secs = oracle.time_sweeps(sampler, Xh, K, cpu_sweeps, cb, 1000, threads)
            alloc_s = threads * rows * cpu_sweeps / secs
            result["cpu_baseline"] = {
                "value": alloc_s / N, "unit": "sweeps/s", "cores": threads, "kind": "port",
                "sample": "%d independent chains (one per thread) x %d sweeps over the first %d rows; "
                          "allocations/s / N" % (threads, cpu_sweeps, rows),
                "allocations_per_s": alloc_s, "seconds": secs}
            result["config"]["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
    ch.close()
    del X, m
    torch.cuda.empty_cache()
    # the other BASELINE points, measured in the same run (single GPU only): configs[1] and the
    # north-star shape.  Reported beside the headline, never instead of it.
    if world == 1 and not args.no_extra and args.workload == "c5" and not (args.n or args.k):
        extra = {}
        if args.x_layout == "bits":
            # the kernel that streams the int32 matrix in place: the HBM-roofline measurement proper
            e = measure("c5", args.steps, args.warmup, args.burn, args.batch, x_layout="int32")
            bps = sweep_bytes(e["N"], e["P"], "int32")
            gbps = bps * args.steps / (e["kern_ms"] * 1e-3) / 1e9
            tr = None
            try:
                tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("c5")
            except Exception:
                pass
            extra["c5_int32"] = {"workload": "c5 with --x-layout int32 (X streamed as R hands it over)",
                                 "sweeps_per_s": args.steps / e["dt"], "ms_per_step": 1e3 * e["dt"] / args.steps,
                                 "batch": e["batch"], "kernel_ms_per_sweep": e["kern_ms"] / args.steps,
                                 "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": gbps / HBM_PEAK_GBS, "traffic": tr,
                                              "algorithmic_bytes_per_sweep": bps,
                                              "frac_of_copy_kernel": (gbps / copy_gbps) if copy_gbps else None}}
            e["chain"].close()
            del e
            torch.cuda.empty_cache()
        for w, steps in (("c2", 200), ("ns", 50)):
            e = measure(w, steps, 5, args.burn, 0, x_layout=args.x_layout)
            bps = sweep_bytes(e["N"], e["P"], e["layout"])
            extra[w] = {"workload": "gibbs_%s K=%d N=%d P=%d" % (e["sampler"], e["K"], e["N"], e["P"]),
                        "sweeps_per_s": steps / e["dt"], "ms_per_step": 1e3 * e["dt"] / steps,
                        "batch": e["batch"], "kernel_ms_per_sweep": e["kern_ms"] / steps,
                        "algorithmic_GBps": bps * steps / (e["kern_ms"] * 1e-3) / 1e9,
                        "note": "working set %.0f MB: cache-resident, not an HBM measurement" % (bps / 1e6)}
            e["chain"].close()
        result["other_workloads"] = extra
    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()
    return result


def bench_sharded(args, world, rank, local, dev, barrier):
    """One chain over all ranks: every rank builds the same synthetic
    matrix from the same seed and keeps its slice of rows; value = sweeps/s of that single chain."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from bmm_mcmc_amd import multi, synth
    sampler, K, K_true, N, P, dseed = synth.WORKLOADS[args.workload]
    if sampler not in ("stickbreaking", "full"):
        raise SystemExit("--shard needs a stick-breaking or full workload (e.g. --workload c4)")
    if args.n:
        N = args.n
    X, _ = synth.device_matrix(N, P, K_true, dseed, dev)      # same seed => same matrix on every rank
    lo, hi = rank * N // world, (rank + 1) * N // world
    Xl = X[:, lo:hi].contiguous()
    del X
    torch.cuda.empty_cache()
    rng = np.random.default_rng(1000)
    pi0 = np.exp(rng.random(K))
    ch = multi.ShardedChain(sampler, Xl, N, lo, K, pi0 / pi0.sum(), rng.random((K, P)), seed=1000, device=local)
    for _ in range(args.burn + args.warmup):
        ch.sweep()
    ch.chain.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ch.sweep()
    ch.chain.sync()
    barrier()
    dt = multi.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({
            "metric": "gibbs_sweeps_per_s", "value": args.steps / dt, "unit": "sweeps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: gibbs_%s K=%d N=%d P=%d, ONE chain sharded over %d GPU(s)" % (
                args.workload, sampler, K, N, P, world), "rows_per_rank": hi - lo,
                "collective": "all-reduce of %d int32 per sweep" % (K * (P + 1))}}))
    ch.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
