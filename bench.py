#!/usr/bin/env python3
"""Gibbs sweeps/s of the cluster-allocation path on N MI355X GPUs (one chain per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one Gibbs sweep: every z_n resampled once, sufficient statistics and
parameters refreshed.  Default workload "c5" is BASELINE.json's HBM-roofline
configuration, one chain per GPU: gibbs_collapsed, K=20, N=1e7, P=100, synthetic.
The data matrix is generated in HBM on rank 0 and handed to that rank's chain, which packs
it into bit planes; the planes (160 MB, not the 4 GB int32 matrix) are broadcast once over
RCCL -- the only collective on this path; chains are independent, so scaling is weak.

Prints ONE JSON line on rank 0.
  roofline       the dominant kernel, k_resample, timed with HIP events on the chain's own stream
                 inside the timed region.  `achieved` / `peak` / `frac` are SURVEY.md section 8d's figure:
                 algorithmic bytes of the layout actually streamed (bit planes: 4*ceil(P/32) + 8 bytes per
                 observation) per second of kernel time, against the 8 TB/s HBM peak.  With X in bit planes
                 that fraction is small BY DESIGN (the kernel streams 17x fewer bytes than the int32
                 layout) and the kernel is bound by what it does per observation on the CU: `bound` says
                 "valu+lds (compute)", and `valu_util` (VALU wave-instructions issued / issue slots, from
                 the rocprofv3 SQ_INSTS_VALU pass under profiles/ and the launch time measured here),
                 `lds_busy`, `lds_conflict_frac` (PMC passes under profiles/) say how busy the two
                 co-binding pipes are -- utilisation of the instructions this kernel executes, not a
                 fraction of algorithmic work.
  roofline_int32 the same workload on the kernel that streams the int32 matrix as R hands it over
                 (4P + 8 bytes per observation): that one IS HBM-bound (`bound` "hbm") and carries the
                 north star's ">= 40 % of the HBM roofline on the z-resample kernel" claim; the default
                 bit-plane kernel is 1.7-1.8x faster in sweeps/s and is what `value` reports.
  cpu_baseline   the oracle's sufficient-statistics chain (same batch semantics) on the host cores
                 of this box: one chain per thread on as many threads as this process may use, up
                 to one socket's cores; on a bounded row sample, scaled to sweeps/s at the
                 workload's N.  `literal_1thread` is the reference algorithm as written (member-list
                 recomputation, O(N^2 P) per sweep), timed at small N and EXTRAPOLATED.
  other_workloads  the other BASELINE configurations and the north-star shape in the same run
                 (single GPU only): c2, c3, c4, ns (each with its own cpu_baseline), four chains sharing
                 one GPU, the PCIe-inclusive drop-in call at the north-star shape and at C5
                 (ns_end_to_end, c5_end_to_end: host matrix in, S x N trace out, with the library's own
                 phase clock), the rate from a random start.
  multi_gpu      (N > 1) the launcher, the ranks the collective saw, the plane broadcast's milliseconds and
                 every chain's own sweeps/s, so that a scaling record can be checked against its parts.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
# full-rate VALU issue: 256 CUs x 4 SIMDs, one wave-instruction per 4 cycles at 2.4 GHz
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4


def cgroup_cpus():
    """CPU quota of this process's cgroup in whole CPUs (cgroup v2 cpu.max / v1 cfs quota), or None"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // p)
    except Exception:
        pass
    return None


def socket_cores():
    """(cores of one socket, CPUs this process may actually use: affinity mask capped by the cgroup's CPU
    quota and by BMM_BENCH_CPUS -- a one-GPU box of this pool grants 16 of the host's CPUs)"""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cgroup_cpus()
    if quota:
        avail = min(avail, quota)
    if os.environ.get("BMM_BENCH_CPUS"):
        avail = min(avail, int(os.environ["BMM_BENCH_CPUS"]))
    cores = None
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            if line.lower().startswith("core(s) per socket"):
                cores = int(line.split(":")[1])
    except Exception:
        pass
    return cores or avail, avail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--burn", type=int, default=30,
                    help="untimed set-up sweeps before the warm-up, so that the timed region is the "
                         "chain's steady state and not its first sweeps from a random allocation "
                         "(those are reported as from_random_start)")
    ap.add_argument("--workload", default="c5", choices=["c2", "c3", "c4", "c5", "ns"])
    ap.add_argument("--batch", type=int, default=0, help="observations per frozen-statistics batch (0 = default)")
    ap.add_argument("--rows", type=int, default=0, help="override N (debug)")
    ap.add_argument("--k", type=int, default=0, help="override K (debug)")
    ap.add_argument("--x-layout", default="bits", choices=["bits", "int32"],
                    help="what the resample kernel streams: bit planes packed once at hand-over (default) "
                         "or the int32 matrix as R hands it over")
    ap.add_argument("--chains-per-gpu", type=int, default=1,
                    help="resident chains per GPU over one copy of the bit planes, each on its own stream")
    ap.add_argument("--event-stride", type=int, default=4, help="time the resample launches of every n-th sweep")
    ap.add_argument("--no-events", action="store_true",
                    help="debug: no HIP events around the resample launches (roofline is then null)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (other_workloads)")
    ap.add_argument("--shard", action="store_true",
                    help="ONE chain whose rows are split over the ranks (stick-breaking / full workloads "
                         "only; strong scaling, one all-reduce of the statistics per sweep)")
    ap.add_argument("--cpu-rows", type=int, default=1_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU time of a baseline leg")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import bmm_mcmc_amd as bm
    from bmm_mcmc_amd import multi, synth

    world, rank, local = multi.world()
    if world == 1 and args.gpus > 1:
        # not under torch.distributed.run: the native launcher -- one process, one chain and one host
        # thread per GPU, the bit planes broadcast with RCCL inside the library (bmm_chains_broadcast_planes)
        return bench_native(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # BMM_BENCH_REHEARSE=1: every rank on device 0 over gloo -- how the N > 1 code path (plane broadcast,
    # per-rank seeds, max over ranks) is rehearsed on a one-GPU box; never a measurement
    rehearse = os.environ.get("BMM_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rehearse:
        multi.init("gloo")
    else:
        multi.init("nccl", device=dev)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def profile_info(key):
        """per-workload constants measured with rocprofv3 (profiles/traffic.json): HBM bytes per launch
        from the PMC counters, VALU wave-instructions per 64 observations"""
        try:
            v = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
            return v if isinstance(v, dict) else None
        except Exception:
            return None

    def measure(workload, steps, warmup, burn, batch_arg, n_override=0, k_override=0, x_layout="bits", nchains=1,
                random_start=0):
        """`nchains` chains per rank on `workload`; timing of `steps` sweeps of every chain (max over ranks)."""
        sampler, K, K_true, N, P, dseed = synth.WORKLOADS[workload]
        if n_override:
            N = n_override
        if k_override:
            K = k_override
            K_true = min(K_true, K)
        chains = [bm.Chain(sampler, N, P, K, alpha=None, beta=0.5, gamma=0.5, a=1, b=1,
                           batch=batch_arg if batch_arg > 0 else None,
                           seed=multi.chain_seed(1000, rank * nchains + c),  # chain seeds 1000 + c
                           device=local, x_layout=x_layout) for c in range(nchains)]
        multi.LOCAL_DEVICE = local
        ch = chains[0]
        # data: generated in HBM on rank 0 and handed to its chain; what is broadcast is the packed planes
        X = None
        if rank == 0:
            X, _ = synth.device_matrix(N, P, K_true, dseed, dev)
        t_b0 = time.perf_counter()
        if x_layout == "bits":
            multi.broadcast_planes(ch, X, src=0)
            for c in chains[1:]:
                c.share_data(ch)
        else:
            if rank != 0:
                X = torch.empty((P, N), dtype=torch.int32, device=dev)
            multi.broadcast_data(X, src=0)
            for c in chains:
                c.set_data_device(X.data_ptr(), keepalive=X)
        bcast_ms = 1e3 * (time.perf_counter() - t_b0)  # rank 0: pack + broadcast; others: receive
        for ci, c in enumerate(chains):
            rng = np.random.default_rng(multi.chain_seed(1000, rank * nchains + ci))
            if sampler == "collapsed":
                c.set_initial_labels(rng.integers(1, K + 1, N).astype(np.int32))
            elif sampler in ("stickbreaking", "full"):
                pi0 = np.exp(rng.random(K))
                c.set_initial_params(pi0 / pi0.sum(), rng.random((K, P)))
        first = None
        if random_start:  # the first sweeps from a random allocation (more movers, every cell flushed)
            ch.sweeps(0)
            ch.sync()
            t0 = time.perf_counter()
            ch.sweeps(random_start)
            ch.sync()
            first = random_start / (time.perf_counter() - t0)
            burn = max(0, burn - random_start)

        def run(n):
            if nchains == 1:
                ch.sweeps(n)
            else:
                bm.sweep_chains(chains, n)

        run(burn)
        run(warmup)
        for c in chains:
            c.sync()
        # HIP events around the resample launches of every 4th timed sweep: the per-launch average
        # is what the roofline needs, and 16 event records per sweep would cost ~7 % of the sweep
        ch.profile(0 if args.no_events else args.event_stride)
        barrier()
        t0 = time.perf_counter()
        run(steps)
        for c in chains:
            c.sync()
        t_own = time.perf_counter() - t0  # this rank's chains alone, before waiting for the others
        barrier()
        t1 = time.perf_counter()
        kern_ms, kern_n = ch.profile_read()
        ch.profile(False)
        # launches per sweep are structural: one per batch (one in all for the explicit-parameter samplers)
        lps = 1 if sampler in ("stickbreaking", "full") else -(-N // ch.batch)
        per_launch = (kern_ms / kern_n) if kern_n else 0.0
        kern_ms = multi.max_over_ranks(per_launch * lps * steps)  # resample-kernel ms over the timed sweeps
        return {"sampler": sampler, "K": K, "N": N, "P": P, "batch": ch.batch, "shape": ch.kernel_shape(),
                "dt": multi.max_over_ranks(t1 - t0), "kern_ms": kern_ms, "kern_n": lps * steps, "lps": lps,
                "X": X, "chains": chains, "layout": ch.x_layout(), "first": first, "nchains": nchains,
                "bcast_ms": multi.max_over_ranks(bcast_ms),
                # collectives: every rank calls them (here, not under `if rank == 0`)
                "rank_sweeps_per_s": multi.gather_summaries([nchains * steps / t_own])[:, 0].tolist(),
                "ranks_seen": int(round(float(multi.gather_summaries([1.0]).sum())))}

    def close(m):
        for c in m["chains"][1:] + m["chains"][:1]:  # borrowers of the planes first
            c.close()
        m["X"] = None
        torch.cuda.empty_cache()

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

    def roofline_of(m, steps, key):
        """roofline object of one measurement (see the module docstring)"""
        N, P, layout, kern_ms = m["N"], m["P"], m["layout"], m["kern_ms"]
        if not kern_ms:
            return None
        secs = kern_ms * 1e-3
        bps = sweep_bytes(N, P, layout)
        gbps = bps * steps / secs / 1e9
        info = profile_info(key) or {}
        out = {"kernel": "k_resample", "kernel_ms_per_sweep": kern_ms / steps, "launches_per_sweep": m["lps"],
               "lds_bytes": m["shape"]["lds_bytes"], "threads": m["shape"]["threads"],
               "algorithmic_bytes_per_sweep": bps, "traffic": info.get("hbm_bytes_per_launch")}
        if layout == "int32":
            out.update({"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbps / HBM_PEAK_GBS})
            return out
        vpw = info.get("valu_inst_per_64_obs")
        # LDS traffic the scoring cannot avoid: K*G group-table reads (G groups of the shape's width) + Gm
        # own-cluster reads (groups of 3) + K reads of the exponential's table, 8 bytes each, per observation;
        # peak 256 B/clk/CU x 256 CUs x 2.4 GHz
        from bmm_mcmc_amd import _capi
        code = {"collapsed": 0, "dp": 1, "stickbreaking": 2, "full": 3}[m["sampler"]]
        gw = _capi.lib().bmm_spec_group_width_for(code, m["K"], P)
        G = (P + gw - 1) // gw
        own = 0 if m["sampler"] in ("stickbreaking", "full") else (P + _capi.lib().bmm_spec_group_width_own() - 1) // _capi.lib().bmm_spec_group_width_own()
        cats = m["K"] + (1 if m["sampler"] == "dp" else 0)
        lds_tbps = (cats * G + own + cats) * 8 * N * steps / secs / 1e12
        lds_peak = 256 * 256 * 2.4e9 / 1e12
        # frac = SURVEY.md 8d's HBM fraction for the bytes this layout streams; the pipes that bind the kernel
        # are reported beside it
        out.update({"bound": "valu+lds (compute)", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbps / HBM_PEAK_GBS,
                    "lds_read_TBps": lds_tbps, "lds_read_frac_of_peak": lds_tbps / lds_peak, "group_width": gw,
                    "layout": "bit planes: %d bytes per observation and sweep" % (bps // N)})
        if vpw:
            ginst = vpw * (N / 64.0) * steps / secs / 1e9
            out.update({"valu_util": ginst / VALU_PEAK_GINST, "valu_G_wave_inst_per_s": ginst,
                        "valu_peak_G_wave_inst_per_s": VALU_PEAK_GINST, "valu_inst_per_64_obs": vpw,
                        "counters_source": info.get("source")})
        for k in ("lds_busy", "lds_conflict_frac", "valu_busy"):
            if info.get(k) is not None:
                out[k] = info[k]
        out["note"] = ("frac is SURVEY.md 8d's HBM fraction for the layout streamed: small by design -- with X in bit "
                       "planes the kernel streams 17x fewer bytes than the int32 layout and is bound by what it does "
                       "per observation on the CU (K*ceil(P/5) LDS table reads + fp64 adds, K exponentials).  "
                       "valu_util = VALU wave-instructions per launch (SQ_INSTS_VALU, profiles/) / launch time "
                       "measured here / (1024 SIMDs x 2.4 GHz / 4 cycles): issue-slot utilisation of the instructions "
                       "this kernel executes, not a fraction of algorithmic work.  lds_busy / lds_conflict_frac / "
                       "valu_busy are the PMC passes' figures (profiles/): the two pipes are co-binding.  The "
                       "HBM-bound kernel of the same workload is roofline_int32")
        return out

    def cpu_leg(sampler, Xdev, K, N, batch, label):
        """the oracle's sufficient-statistics chain on this box's cores; literal form extrapolated"""
        from oracle import oracle
        per_socket, avail = socket_cores()
        rows = min(args.cpu_rows, N)
        Xh = np.asfortranarray(Xdev[:, :rows].t().cpu().numpy())  # first `rows` (shuffled) rows
        cb = max(1, min(batch, rows))
        # one socket's cores where the box grants them; a one-GPU box of this pool shares its host, and its
        # CPU share (16) can be below what the affinity mask shows -- so the thread count is the better of
        # {16, one socket} by a one-sweep probe on a fifth of the sample, both probes reported
        cand = sorted({max(1, min(16, avail)), max(1, min(per_socket, avail))})
        probes = {}
        Xp = np.asfortranarray(Xh[:max(1000, rows // 5)])
        for t in cand:
            probes[t] = t * Xp.shape[0] / oracle.time_sweeps(sampler, Xp, K, 1, max(1, min(cb, Xp.shape[0])), 1000, t)
        threads = max(probes, key=probes.get)
        # BASELINE.md (ii) also asks for the one-thread rate of the port: two sweeps of the probe sample
        one_thread = Xp.shape[0] * 2 / oracle.time_sweeps(sampler, Xp, K, 2, max(1, min(cb, Xp.shape[0])), 1000, 1)
        probe = oracle.time_sweeps(sampler, Xh, K, 1, cb, 1000, threads)  # size the leg to ~cpu-seconds
        cpu_sweeps = int(max(2, min(200, round(args.cpu_seconds / max(probe, 1e-3)))))
        secs = oracle.time_sweeps(sampler, Xh, K, cpu_sweeps, cb, 1000, threads)
        alloc_s = threads * rows * cpu_sweeps / secs
        out = {"value": alloc_s / N, "unit": "sweeps/s", "cores": threads, "kind": "port",
               "sample": "%s: %d independent chains (one per thread) x %d sweeps over the first %d rows; "
                         "allocations/s / N" % (label, threads, cpu_sweeps, rows),
               "allocations_per_s": alloc_s, "seconds": secs,
               "cores_per_socket": per_socket, "cpus_available_to_this_process": avail,
               "thread_count_probe_allocations_per_s": {str(t): v for t, v in probes.items()},
               "one_thread": {"value": one_thread / N, "unit": "sweeps/s", "cores": 1, "allocations_per_s": one_thread}}
        if threads < per_socket:
            out["one_socket_extrapolated"] = {
                "value": alloc_s / N * per_socket / threads,
                "note": "EXTRAPOLATION: this process may use %d of the socket's %d cores; measured rate x %d/%d, "
                        "assuming the port scales ideally" % (avail, per_socket, per_socket, threads)}
        if sampler == "collapsed":
            # the reference algorithm as written, one thread: c * N^2 * P per sweep, fitted at small N
            P = Xh.shape[1]
            pts = []
            for n in (1000, 2000):
                Xs = np.asfortranarray(Xh[:n])
                z0 = np.random.default_rng(0).integers(1, K + 1, n).astype(np.int32)
                t0 = time.perf_counter()
                oracle.collapsed(Xs, z0, 3, K, 1.0, 0.5, 0.5, 1, 1, 0, seed=1, literal=True)
                pts.append((n, (time.perf_counter() - t0) / 2))
            c = float(np.mean([t / (n * n * P) for n, t in pts]))
            out["literal_1thread"] = {
                "value": 1.0 / (c * N * N * P), "unit": "sweeps/s", "cores": 1, "kind": "port",
                "note": "EXTRAPOLATION of the reference algorithm as written (member lists recomputed per "
                        "(i, k, d), src/collapsed_gibbs.cpp:99-129): c * N^2 * P with c = %.3g s fitted at "
                        "N = 1000 and 2000 (%.3f s, %.3f s per sweep)" % (c, pts[0][1], pts[1][1])}
        return out

    def end_to_end(Xh, K, nsamples, burnin, label):
        """the drop-in call itself, host matrix in and S x N trace out, PCIe included (never `value`): five calls
        -- the first also pays for pinned staging, device blocks and a stream that the library keeps for later
        calls -- reported as the median of the last four, every call listed, with the library's own phase clock
        of the last one"""
        import ctypes as C
        from bmm_mcmc_amd import _capi
        Nn = Xh.shape[0]
        z0 = np.random.default_rng(0).integers(1, K + 1, Nn).astype(np.int32)
        times, out = [], None
        for _ in range(5):
            del out  # outside the clock: releasing the previous call's S x N matrix is the caller's business
            t0 = time.perf_counter()
            out = bm.gibbs_collapsed(Xh, nsamples, K, burnin=burnin, seed=1, initial_K=z0)
            times.append(time.perf_counter() - t0)
        later = float(np.median(times[1:]))
        ms = (C.c_double * 6)()
        _capi.lib().bmm_last_run_phases(ms)
        names = ("pack_left_and_upload", "create_and_start_state", "enqueue", "device_wait", "trace_out", "release")
        return {"workload": label, "sweeps_per_s": nsamples / later, "seconds": later, "first_call_seconds": times[0],
                "calls_seconds": [round(t, 5) for t in times],
                "kept_sweeps": int(out["z"].shape[0]), "host_threads": int(_capi.lib().bmm_host_threads()),
                "phases_ms": {n: round(v, 3) for n, v in zip(names, ms)},
                "note": "through the Python mirror of the R wrapper; X is validated and packed into bit planes by the "
                        "host's cores while the chain is created, only the planes cross PCIe; the label trace leaves "
                        "in blocks, one byte per label, widened on the host"}

    if args.shard:
        return bench_sharded(args, world, rank, local, dev, barrier)
    rs = 3 if (args.workload in ("c5", "ns", "c2") and args.chains_per_gpu == 1 and args.x_layout == "bits") else 0
    m = measure(args.workload, args.steps, args.warmup, args.burn, args.batch, args.rows, args.k, args.x_layout,
                nchains=args.chains_per_gpu, random_start=rs)
    sampler, K, N, P, batch = m["sampler"], m["K"], m["N"], m["P"], m["batch"]
    dt, nch = m["dt"], m["nchains"]

    result = None
    if rank == 0:
        layout = m["layout"]
        key = args.workload + ("" if layout == "int32" else "_bits")
        result = {
            "metric": "gibbs_sweeps_per_s",
            "value": world * nch * args.steps / dt,
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
            "config": {"workload": "%s: gibbs_%s K=%d N=%d P=%d, %d chain%s per GPU" % (
                           args.workload, sampler, K, N, P, nch, "" if nch == 1 else "s"),
                       "sampler": sampler, "K": K, "N": N, "P": P, "batch": batch, "chains": world * nch,
                       "chains_per_gpu": nch, "burn_sweeps": args.burn,
                       "x_layout": ("bit planes, ceil(P/32) words per observation, packed once at hand-over"
                                    if layout == "bits" else "int32 column-major (as R hands it over)"),
                       "allocations_per_s": world * nch * args.steps * N / dt},
            "roofline": roofline_of(m, args.steps, key),
        }
        if world > 1:
            seen = m["ranks_seen"]
            result["multi_gpu"] = {
                "launcher": "torch.distributed.run: one process per GPU; the bit planes broadcast once from rank 0 "
                            "(torch.distributed.broadcast, backend %s)" % ("gloo (REHEARSAL on one device, not a measurement)"
                                                                          if rehearse else "nccl = RCCL"),
                "ranks_seen_by_the_collective": seen, "world_size": world,
                "plane_broadcast_ms": m["bcast_ms"],
                "plane_bytes": 4 * ((P + 31) // 32) * N,
                "per_rank_sweeps_per_s": m["rank_sweeps_per_s"],
                "note": "value = all chains' sweeps / the slowest rank's time (barrier on both sides); per_rank_* is each "
                        "rank's own clock around its own chains"}
        if m["first"]:
            result["from_random_start"] = {
                "sweeps_per_s": m["first"],
                "note": "the first 3 sweeps of the chain from a uniformly random allocation (every observation "
                        "moves, every statistic cell is flushed); `value` is the steady state after %d sweeps" % args.burn}
        copy_gbps = copy_rate() if world == 1 else None
        if copy_gbps and result["roofline"]:
            result["roofline"]["copy_kernel_GBps"] = copy_gbps
        if not args.no_cpu and world == 1 and m["X"] is not None:
            result["cpu_baseline"] = cpu_leg(sampler, m["X"], K, N, batch, args.workload)
            result["config"]["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
    c5_e2e = None
    if (world == 1 and not args.no_extra and args.workload == "c5" and not (args.rows or args.k) and args.chains_per_gpu == 1
            and m["X"] is not None and sampler == "collapsed"):
        # C5 through the drop-in call: the same matrix, copied to the host (N x P column-major as R holds it)
        Xh = m["X"].t().cpu().numpy()
        c5_e2e = end_to_end(Xh, K, 30, 3, "gibbs_collapsed(host X 4 GB in, 30 sweeps, 27 kept: S x N trace 1.08 GB out) at C5, PCIe included")
        del Xh
    close(m)
    del m
    # the other BASELINE points, measured in the same run (single GPU only).  Reported beside the
    # headline, never instead of it.
    if world == 1 and not args.no_extra and args.workload == "c5" and not (args.rows or args.k) and args.chains_per_gpu == 1:
        extra = {}
        if args.x_layout == "bits":
            # the kernel that streams the int32 matrix in place: the HBM-roofline measurement proper
            e = measure("c5", args.steps, args.warmup, args.burn, args.batch, x_layout="int32")
            r = roofline_of(e, args.steps, "c5")
            if r and copy_gbps:
                r["frac_of_copy_kernel"] = r["achieved"] / copy_gbps
            extra["c5_int32"] = {"workload": "c5 with --x-layout int32 (X streamed as R hands it over)",
                                 "sweeps_per_s": args.steps / e["dt"], "ms_per_step": 1e3 * e["dt"] / args.steps,
                                 "batch": e["batch"], "roofline": r}
            if r:
                result["roofline_int32"] = dict(r, sweeps_per_s=args.steps / e["dt"],
                                                note="the same workload on the kernel that streams the int32 matrix in place "
                                                     "(bmm_chain_set_x_layout BMM_X_INT32): HBM-bound, carries the north star's "
                                                     ">= 40 % of HBM roofline claim; the default bit-plane kernel is faster in "
                                                     "sweeps/s and is what `value` reports")
            close(e)
            del e
        for w, steps in (("c2", 200), ("c3", 50), ("c4", 50), ("ns", 50)):
            e = measure(w, steps, 5, args.burn, 0, x_layout=args.x_layout, random_start=3 if w == "ns" else 0)
            bps = sweep_bytes(e["N"], e["P"], e["layout"])
            extra[w] = {"workload": "gibbs_%s K=%d N=%d P=%d" % (e["sampler"], e["K"], e["N"], e["P"]),
                        "sweeps_per_s": steps / e["dt"], "ms_per_step": 1e3 * e["dt"] / steps,
                        "batch": e["batch"], "launches_per_sweep": e["lps"],
                        "kernel_ms_per_sweep": e["kern_ms"] / steps,
                        "allocations_per_s": steps * e["N"] / e["dt"],
                        "algorithmic_GBps": bps * steps / (e["kern_ms"] * 1e-3) / 1e9 if e["kern_ms"] else None,
                        "note": "working set %.0f MB: cache-resident, not an HBM measurement" % (bps / 1e6)}
            if w == "ns" and e["first"]:
                extra[w]["from_random_start_sweeps_per_s"] = e["first"]
            if not args.no_cpu and e["X"] is not None:
                save = args.cpu_seconds
                if w != "ns":
                    args.cpu_seconds = min(save, 4.0)  # bounded: the default run must finish within minutes
                extra[w]["cpu_baseline"] = cpu_leg(e["sampler"], e["X"], e["K"], e["N"], e["batch"], w)
                args.cpu_seconds = save
                extra[w]["gpu_over_cpu"] = extra[w]["sweeps_per_s"] / extra[w]["cpu_baseline"]["value"]
            close(e)
            del e
        # MCMC practice runs several chains: four resident chains on this GPU over one copy of the planes
        e = measure("ns", 50, 5, args.burn, 0, x_layout="bits", nchains=4)
        extra["ns_4chains"] = {"workload": "north-star shape, 4 chains on one GPU (bmm_chain_share_data, "
                                           "bmm_chains_sweeps: one stream and one host thread per chain)",
                               "chains_per_gpu": 4, "sweeps_per_s": 4 * 50 / e["dt"],
                               "ms_per_step_per_chain": 1e3 * e["dt"] / 50,
                               "vs_single_chain": (4 * 50 / e["dt"]) / extra["ns"]["sweeps_per_s"]}
        close(e)
        del e
        # the drop-in entry point end to end: host matrix in over PCIe, S x N trace out (never `value`)
        Xh, _, _, _ = synth.host_matrix(1_000_000, 50, 20, 22)
        extra["ns_end_to_end"] = end_to_end(Xh, 20, 220, 200, "gibbs_collapsed(host X 200 MB in, 220 sweeps, 20 kept: S x N trace "
                                                              "80 MB out) at the north-star shape, PCIe included")
        extra["ns_end_to_end"]["vs_resident"] = extra["ns_end_to_end"]["sweeps_per_s"] / extra["ns"]["sweeps_per_s"]
        del Xh
        if c5_e2e:
            c5_e2e["vs_resident"] = c5_e2e["sweeps_per_s"] / result["value"]
            extra["c5_end_to_end"] = c5_e2e
        result["other_workloads"] = extra
    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()
    return result


def bench_native(args):
    """`python bench.py --gpus N` without torch.distributed.run: the same measurement from one process.
    One resident chain per GPU (seeds 1000 + c), the data generated and packed on GPU 0, its bit planes
    broadcast once over RCCL by the library itself, the sweeps of all chains enqueued by one host thread
    per chain (bmm_chains_sweeps); value = sweeps of all chains / wall time of the timed region."""
    import numpy as np
    import torch
    import bmm_mcmc_amd as bm
    from bmm_mcmc_amd import synth
    n = args.gpus
    if torch.cuda.device_count() < n:
        raise SystemExit("%d GPUs asked for, %d visible" % (n, torch.cuda.device_count()))
    sampler, K, K_true, N, P, dseed = synth.WORKLOADS[args.workload]
    if args.rows:
        N = args.rows
    chains = [bm.Chain(sampler, N, P, K, batch=args.batch if args.batch > 0 else None, seed=1000 + c, device=c)
              for c in range(n)]
    X, _ = synth.device_matrix(N, P, K_true, dseed, torch.device("cuda", 0))
    chains[0].set_data_device(X.data_ptr())
    del X
    t_b = time.perf_counter()
    bm.broadcast_planes(chains)
    bcast_ms = 1e3 * (time.perf_counter() - t_b)
    for c, ch in enumerate(chains):
        rng = np.random.default_rng(1000 + c)
        if sampler == "collapsed":
            ch.set_initial_labels(rng.integers(1, K + 1, N).astype(np.int32))
        elif sampler in ("stickbreaking", "full"):
            pi0 = np.exp(rng.random(K))
            ch.set_initial_params(pi0 / pi0.sum(), rng.random((K, P)))
    bm.sweep_chains(chains, args.burn + args.warmup)
    for ch in chains:
        ch.sync()
    t0 = time.perf_counter()
    bm.sweep_chains(chains, args.steps)
    for ch in chains:
        ch.sync()
    dt = time.perf_counter() - t0
    print(json.dumps({
        "metric": "gibbs_sweeps_per_s", "value": n * args.steps / dt, "unit": "sweeps/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: gibbs_%s K=%d N=%d P=%d, 1 chain per GPU" % (args.workload, sampler, K, N, P),
                   "sampler": sampler, "K": K, "N": N, "P": P, "batch": chains[0].batch, "chains": n,
                   "launcher": "native: one process, one host thread per GPU, RCCL broadcast of the bit planes "
                               "inside the library (bmm_chains_broadcast_planes)"},
        "multi_gpu": {"launcher": "native (bmm_chains_broadcast_planes: ncclCommInitAll + one ncclBroadcast; bmm_chains_sweeps)",
                      "devices": n, "plane_broadcast_ms": bcast_ms, "plane_bytes": 4 * ((P + 31) // 32) * N}}))
    for ch in chains:
        ch.close()


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
    if args.rows:
        N = args.rows
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
