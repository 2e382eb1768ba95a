"""Build recipe for the HIP library (gfx950 only).  hipcc cross-compiles without a GPU.

Two variants of the same sources:
  lib/libbmmmcmc_hip.so      the product: reads no environment, no debug hooks
  lib/libbmmmcmc_hip_dbg.so  -DBMM_DEBUG_HOOKS: the kernel-steering environment switches the parity
                             tests use to reach every kernel variant (BMM_DEBUG_GENERIC,
                             BMM_DEBUG_THREADS, BMM_DEBUG_NOSPLIT, BMM_X_LAYOUT_INT32)
`BMM_LIB_PATH` makes _capi.py load another build (tools/exp_lib.sh, tools/diag.sh) -- nothing ever
overwrites the product library in place.
"""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "chain.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "kernels.hip.h"), os.path.join(HERE, "csrc", "bmm_spec.h"),
        os.path.join(HERE, "csrc", "bmm_exp256.h"), os.path.join(HERE, "csrc", "host_crew.h"),
        os.path.join(os.path.dirname(HERE), "include", "bmm_mcmc.h")]
LIB = os.path.join(HERE, "lib", "libbmmmcmc_hip.so")
LIB_DBG = os.path.join(HERE, "lib", "libbmmmcmc_hip_dbg.so")

# -ffp-contract=off: the spec arithmetic (csrc/bmm_spec.h) fuses only where it says fma_
# -amdgpu-sched-strategy=iterative-ilp: the resample kernels are VALU-issue-bound with LDS reads to
#   hide; this list scheduler measured +2 % on C5 (both layouts) over the default, same results
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm",
         "-amdgpu-sched-strategy=iterative-ilp", "-fPIC", "-shared", "-Wl,-rpath,/opt/rocm/lib"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP library cannot be built")
    return exe


def _fingerprint(extra=()):
    """sha256 over the sources a library is built from and the flags it is built with"""
    h = hashlib.sha256()
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(FLAGS + list(extra)).encode())
    return h.hexdigest()


def _extra(lib):
    return ["-DBMM_DEBUG_HOOKS"] if lib == LIB_DBG else []


def stale(lib=LIB):
    """A library is current when the fingerprint written beside it at build time equals that of the sources
    as they are now: content, not modification times (which a snapshot copied to another machine does not
    keep, and which say nothing about which sources an .so left over from an earlier checkout was built from)."""
    try:
        with open(lib + ".src-sha256") as f:
            return (not os.path.exists(lib)) or f.read().strip() != _fingerprint(_extra(lib))
    except OSError:
        return True


def _cmd(lib, extra):
    return [hipcc()] + FLAGS + list(extra) + ["-o", lib, SRC]


def build(force=False, verbose=False, debug_variant=False):
    """Build the product library (and, with debug_variant, the test variant beside it, in parallel)."""
    jobs = []
    for lib, extra, want in ((LIB, [], True), (LIB_DBG, ["-DBMM_DEBUG_HOOKS"], debug_variant)):
        if want and (force or stale(lib)):
            os.makedirs(os.path.dirname(lib), exist_ok=True)
            cmd = _cmd(lib + ".tmp", extra)
            if verbose:
                print(" ".join(cmd))
            jobs.append((lib, subprocess.Popen(cmd)))
    for lib, proc in jobs:
        if proc.wait() != 0:
            raise RuntimeError("hipcc failed building " + lib)
        os.replace(lib + ".tmp", lib)  # never a half-written library under the product's name
        with open(lib + ".src-sha256", "w") as f:
            f.write(_fingerprint(_extra(lib)) + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True, debug_variant=True))
