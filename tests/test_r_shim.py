"""The R side of the boundary, as far as an image without R can check it:

* r-shim/bmmmcmc_shim.c passes `gcc -fsyntax-only -Wall -Wextra -Werror` against test-only declarations of
  the R API symbols it uses (tests/r_api_stub: declarations, nothing linked or run);
* its registration table carries the reference's seven .Call entries name for name and arity for
  arity.  REFERENCE_TABLE is /root/reference/src/RcppExports.cpp:137-146 committed as constants (the
  reference is not on the GPU box; where it is present the constants are re-read from it);
* the R wrappers pass the *_ex entry points exactly as many arguments as the shim registers.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "bmm-mcmc_amd", "r-shim", "bmmmcmc_shim.c")
WRAPPERS = os.path.join(ROOT, "bmm-mcmc_amd", "R", "gibbs.R")

# src/RcppExports.cpp:137-146 of stulacy/bmm-mcmc
REFERENCE_TABLE = {
    "_bmmmcmc_collapsed_gibbs_cpp": 13,
    "_bmmmcmc_collapsed_gibbs_dp_cpp": 12,
    "_bmmmcmc_rdirichlet_cpp": 1,
    "_bmmmcmc_gibbs_cpp": 14,
    "_bmmmcmc_my_lpsolve": 1,
    "_bmmmcmc_my_stephens_batch": 2,
    "_bmmmcmc_gibbs_stickbreaking_cpp": 14,
}
EX_TABLE = {
    "_bmmmcmc_collapsed_gibbs_ex": 17,
    "_bmmmcmc_collapsed_gibbs_dp_ex": 16,
    "_bmmmcmc_gibbs_ex": 17,
    "_bmmmcmc_gibbs_stickbreaking_ex": 17,
}
ENTRY = re.compile(r'\{"(_bmmmcmc_\w+)",\s*\(DL_FUNC\)\s*&\s*(\w+),\s*(\d+)\}')


def _table(path):
    src = open(path).read()
    body = src[src.index("CallEntries[]"):]
    return {m.group(1): (m.group(2), int(m.group(3))) for m in ENTRY.finditer(body)}


def _definitions(src):
    """name -> number of SEXP parameters of every `SEXP name(...) {` definition in the shim"""
    out = {}
    for m in re.finditer(r"^SEXP (_bmmmcmc_\w+)\(([^)]*)\)\s*\{", src, flags=re.M | re.S):
        out[m.group(1)] = len([a for a in m.group(2).split(",") if a.strip()])
    return out


@pytest.mark.parametrize("defs", [[], ["-DBMM_SHIM_FORWARD"]])
def test_shim_is_valid_c_against_the_r_api_declarations(defs):
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    cmd = [gcc, "-fsyntax-only", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
           "-Wno-cast-function-type",  # the (DL_FUNC) cast of R's own registration idiom
           "-I" + os.path.join(ROOT, "tests", "r_api_stub"), "-I" + os.path.join(ROOT, "include")] + defs + [SHIM]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_shim_object_defines_every_registered_symbol(tmp_path):
    """Compiled to an object (against the same declarations), the shim defines R_init_bmmmcmc and every entry
    it registers; with -DBMM_SHIM_FORWARD the three untouched entry points become undefined references that
    the package's own objects resolve."""
    gcc, nm = shutil.which("gcc"), shutil.which("nm")
    assert gcc and nm
    base = [gcc, "-c", "-std=gnu11", "-fPIC", "-Wno-cast-function-type", "-I" + os.path.join(ROOT, "tests", "r_api_stub"),
            "-I" + os.path.join(ROOT, "include")]
    untouched = {"_bmmmcmc_rdirichlet_cpp", "_bmmmcmc_my_lpsolve", "_bmmmcmc_my_stephens_batch"}
    for defs, undefined in (([], set()), (["-DBMM_SHIM_FORWARD"], untouched)):
        obj = str(tmp_path / ("shim%d.o" % len(defs)))
        subprocess.run(base + defs + ["-o", obj, SHIM], check=True, capture_output=True)
        out = subprocess.run([nm, obj], capture_output=True, text=True, check=True).stdout
        defined = {l.split()[-1] for l in out.splitlines() if " T " in l}
        undef = {l.split()[-1] for l in out.splitlines() if l.strip().startswith("U ")}
        names = set(_table(SHIM))
        assert "R_init_bmmmcmc" in defined
        assert names - undefined <= defined, names - undefined - defined
        assert undefined <= undef
        for sym in ("bmm_collapsed_run", "bmm_dp_run", "bmm_sb_run", "bmm_full_run", "bmm_multi_run", "bmm_last_error"):
            assert sym in undef          # resolved by libbmmmcmc_hip.so, which exports them (test_capi_cpu)


def test_registration_table_matches_the_reference():
    got = _table(SHIM)
    for name, arity in REFERENCE_TABLE.items():
        assert name in got, name + " is not registered"
        assert got[name] == (name, arity), (name, got[name])
    for name, arity in EX_TABLE.items():
        assert got[name] == (name, arity)
    assert got["_bmmmcmc_set_progress"] == ("_bmmmcmc_set_progress", 1)
    assert set(got) == set(REFERENCE_TABLE) | set(EX_TABLE) | {"_bmmmcmc_set_progress"}
    # the order of the reference's rows is kept too (cosmetic, but it makes a diff of the two tables empty)
    assert list(got)[:7] == list(REFERENCE_TABLE)


def test_registered_arity_is_the_defined_arity():
    src = open(SHIM).read()
    defs = _definitions(src)
    for name, (fn, arity) in _table(SHIM).items():
        assert defs[fn] == arity, (name, defs[fn], arity)


def test_constants_are_the_reference_registration_table():
    ref = "/root/reference/src/RcppExports.cpp"
    if not os.path.exists(ref):
        pytest.skip("the reference is not on this machine; REFERENCE_TABLE is its committed copy")
    got = {k: v[1] for k, v in _table(ref).items()}
    assert got == REFERENCE_TABLE


def test_r_wrappers_pass_what_the_shim_registers():
    src = open(WRAPPERS).read()
    for name, arity in EX_TABLE.items():
        m = re.search(r'\.Call\("%s",\s*PACKAGE = "bmmmcmc",(.*?)\)\n' % name, src, flags=re.S)
        assert m, name
        args = [a for a in re.sub(r"\([^()]*\)", "", m.group(1)).split(",") if a.strip()]
        assert len(args) == arity, (name, len(args), arity)
    # and the unchanged R/RcppExports.R of the reference calls the *_cpp names with the reference arities
    ref = "/root/reference/R/RcppExports.R"
    if os.path.exists(ref):
        rsrc = open(ref).read()
        for name, arity in REFERENCE_TABLE.items():
            m = re.search(r"\.Call\('%s', PACKAGE = 'bmmmcmc'(.*?)\)\n" % name, rsrc)
            assert m and len([a for a in m.group(1).split(",") if a.strip()]) == arity
