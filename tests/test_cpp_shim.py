"""include/distributions_hip.hpp: a C++ caller written like a caller of the
reference's headers compiles against the shim (CPU) and, on the GPU, produces
the oracle's sequential chain."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "row_update")


def build():
    subprocess.check_call(
        ["g++", "-std=c++11", "-I" + os.path.join(ROOT, "include"),
         os.path.join(ROOT, "examples", "row_update.cc"),
         "-L" + os.path.join(ROOT, "distributions_amd"),
         "-ldistributions_hip",
         "-Wl,-rpath," + os.path.join(ROOT, "distributions_amd"),
         "-o", EXE])


def test_shim_example_compiles_and_links():
    build()
    assert os.path.exists(EXE)


BENCH_EXE = os.path.join(ROOT, "examples", "mixture_bench")


def build_bench():
    """the reference-shaped benchmark: only the include path names this
    library (include/compat first), no source-level mention of it"""
    src = os.path.join(ROOT, "examples", "mixture_bench.cc")
    text = open(src).read()
    code = "\n".join(line for line in text.splitlines()
                     if not line.lstrip().startswith("//"))
    assert "distributions_hip" not in code and "dist_" not in code
    subprocess.check_call(
        ["g++", "-std=c++11", "-Wall", "-Werror",
         "-I" + os.path.join(ROOT, "include", "compat"), src,
         "-L" + os.path.join(ROOT, "distributions_amd"),
         "-ldistributions_hip",
         "-Wl,-rpath," + os.path.join(ROOT, "distributions_amd"),
         "-o", BENCH_EXE])


def test_reference_shaped_benchmark_compiles_with_the_include_path_changed():
    build_bench()
    assert os.path.exists(BENCH_EXE)


REF_HARNESS = os.path.join(ROOT, "oracle", "_ref", "mixture_cc_on_hip")


@pytest.mark.skipif(not os.path.isdir("/root/reference/benchmarks"),
                    reason="the reference's sources exist in the build "
                           "container only (the binary travels)")
def test_the_references_own_harness_compiles_against_the_shim():
    """benchmarks/mixture.cc itself -- the C++ caller SURVEY 8b names -- with
    only its include path changed: Model::Scorer, Mixture::groups() of a const
    mixture, Shared::EXAMPLE() of all six models, sample_int, vector_zero,
    current_time_us, demangle.  Compiled where it lies (oracle/Makefile)."""
    if os.path.exists(REF_HARNESS):
        os.remove(REF_HARNESS)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"),
                           "_ref/mixture_cc_on_hip"])
    assert os.path.exists(REF_HARNESS)
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only",
                                         REF_HARNESS], text=True)
    assert "dist_scorer_init" in undefined and "dist_mixture_score_value" in undefined


@pytest.mark.gpu
def test_the_references_own_harness_runs_on_the_gpu():
    """... and runs: six models, 1 / 10 / 100 / 1000 groups each, both of its
    columns (per-group Scorers, the Mixture) positive"""
    if not os.path.exists(REF_HARNESS):
        pytest.skip("oracle/_ref/mixture_cc_on_hip was not built")
    out = subprocess.check_output([REF_HARNESS], text=True, timeout=900)
    rows = [x.split() for x in out.splitlines()
            if x and x.split()[0] in ("1", "10", "100", "1000")]
    assert len(rows) == 24, out
    assert all(float(r[1]) > 0 and float(r[2]) > 0 for r in rows)
    assert out.count("(cells/us)") == 6


SCORER_EXE = os.path.join(ROOT, "examples", "scorer_check")


def build_scorer_check():
    subprocess.check_call(
        ["g++", "-std=c++11", "-Wall", "-Werror",
         "-I" + os.path.join(ROOT, "include", "compat"),
         os.path.join(ROOT, "examples", "scorer_check.cc"),
         "-L" + os.path.join(ROOT, "distributions_amd"),
         "-ldistributions_hip",
         "-Wl,-rpath," + os.path.join(ROOT, "distributions_amd"),
         "-o", SCORER_EXE])


def test_scorer_check_compiles():
    build_scorer_check()


@pytest.mark.gpu
def test_scorer_eval_equals_score_value_group():
    """Model::Scorer (init + eval) against the mixture's cached scorer on the
    device, every model: the reference's own tolerance (tests/util.py:42,
    1e-3 relative; test_models.py:537-594), bit-identical where the two are
    the same formula (all but the categorical kinds, whose Scorer divides
    before it takes the logarithm, dd.hpp:238-244 / dpd.hpp:312-333); and
    Scorer.eval == Group.score_value bit for bit (dd.hpp:160-167)."""
    build_scorer_check()
    lines = subprocess.check_output([SCORER_EXE], text=True).splitlines()
    got = {x.split()[0]: x.split() for x in lines}
    assert sorted(got) == ["bb", "bnb", "dd", "dpd", "gp", "nich"]
    for name, f in got.items():
        pairs, same, worst = int(f[2]), int(f[4]), float(f[6])
        assert pairs > 100 and worst < 1e-3, (name, f)
        if name not in ("dd", "dpd"):
            assert same == pairs, (name, f)


@pytest.mark.gpu
def test_reference_shaped_benchmark_runs():
    """benchmarks/mixture.cc:104-115 through the compat headers: every model,
    1 / 10 / 100 groups; the accumulated scores are finite"""
    build_bench()
    lines = subprocess.check_output([BENCH_EXE, "100"], text=True).splitlines()
    sums = [float(x.split()[1]) for x in lines if x.startswith("checksum")]
    assert len(sums) == 5 and all(np.isfinite(sums)) and any(sums)
    rates = [float(x.split()[1]) for x in lines
             if x and x.split()[0] in ("1", "10", "100")]
    assert len(rates) == 15 and min(rates) > 0


@pytest.mark.gpu
def test_shim_example_reproduces_the_sequential_chain():
    build()
    lines = subprocess.check_output([EXE], text=True).splitlines()
    out = lines[0].split()
    groups = int(out[1])
    got = [int(v) for v in out[3:]]
    # the grid's first entry is the plain score_data; the wire round trip holds
    grid = lines[1].split()
    assert grid[0] == "grid" and grid[4] == "single"
    assert abs(float(grid[1]) - float(grid[5])) < 1e-5
    assert len({grid[1], grid[2], grid[3]}) == 3
    assert lines[2].endswith("roundtrip ok")
    values = np.array([0, 1, 0, 2, 0, 1, 0, 3], np.uint32)
    assign = (np.arange(8) % 3).astype(np.uint32)
    m = ol.OracleMixture(1.0, 0.2, [ol.make_shared(ol.DD, alphas=[0.5] * 4)])
    m.init_from_assignments([values], assign, 3, 1)
    st = ol.oracle().orc_rng_seed(1)
    for _ in range(3):
        st = m.gibbs_sequential(0, 8, st)
    assert groups == len(m)
    assert got == list(m.assign)


def test_dpd_shared_stick_breaking_through_the_shim_equals_the_mirror():
    """DirichletProcessDiscrete::Shared::add_value / remove_value / realize
    (dpd.hpp:66-101) called as the reference's C++ callers call them
    (examples/dpd_shared_check.cc) == the same calls through
    distributions_amd.lp.models.dpd: one implementation behind both
    (dist_dpd_shared_*), one engine.  Host-side: no GPU needed."""
    exe = os.path.join(ROOT, "examples", "dpd_shared_check")
    subprocess.check_call(
        ["g++", "-std=c++11", "-Wall", "-Werror",
         "-I" + os.path.join(ROOT, "include", "compat"),
         os.path.join(ROOT, "examples", "dpd_shared_check.cc"),
         "-L" + os.path.join(ROOT, "distributions_amd"),
         "-ldistributions_hip",
         "-Wl,-rpath," + os.path.join(ROOT, "distributions_amd"),
         "-o", exe])
    lines = subprocess.check_output([exe], text=True).splitlines()
    from distributions_amd.lp import random as lprandom
    from distributions_amd.lp.models import dpd
    lprandom.seed(7)
    shared = dpd.Shared.from_dict({'gamma': 2.0, 'alpha': 2.0, 'betas': {},
                                   'counts': {}})
    for value in [5, 4, 3, 2, 1, 0, 3, 2, 1]:
        shared.add_value(value)
    shared.remove_value(5)
    shared.add_value(77)
    size, beta0, slot77, slot4 = lines[1].split()
    assert int(size) == len(shared.dump()['betas'])
    assert np.float32(float(beta0)) == np.float32(shared.beta0)
    assert (int(slot77), int(slot4)) == (shared.remap(77), shared.remap(4))
    got = np.array([float(x) for x in lines[2].split()], np.float32)
    assert np.array_equal(got, shared.params.betas)
    shared.realize()
    size, beta0, state, dim = lines[0].split()
    assert int(size) == len(shared.dump()['betas']) and float(beta0) == 0.0
    assert int(state) == lprandom.get_rng().state
    assert int(dim) == shared.params.dim
    assert lines[3].split() == ["100", "0.5", "0", "42"]
