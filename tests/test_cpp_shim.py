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
