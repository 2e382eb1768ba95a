"""The host threads of a *_run call (bmm-mcmc_amd/csrc/host_crew.h: X is packed by them on the way in, the label
trace widened on the way out) on their own, without a device: every index of every job exactly once, asynchronous
and blocking jobs, crews that come and go -- once plainly, once under ThreadSanitizer (data races and lock-order
problems in the hand-over of jobs would otherwise show up as rare wrong traces on the GPU box only)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_crew", "crew_check.cpp")
INC = os.path.join(ROOT, "bmm-mcmc_amd", "csrc")


@pytest.mark.parametrize("flags", [("-O2",), ("-O1", "-g", "-fsanitize=thread")], ids=["plain", "tsan"])
def test_host_crew_jobs(tmp_path, flags):
    exe = str(tmp_path / "crew_check")
    subprocess.run(["g++", "-std=c++17", "-pthread", *flags, "-I", INC, SRC, "-o", exe], check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    sys.stderr.write(r.stderr[-4000:])
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout, r.stderr[-2000:])
