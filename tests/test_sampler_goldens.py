"""The samplers against the REFERENCE's own softmax, at the distribution level
(SURVEY 8(c) golden set (3)): the probabilities come
from distributions/util.py:33-38 scores_to_probs, run where it lies by
tests/golden/make_sampler_goldens.py.  The oracle's sample_from_scores_overwrite
(random.cc:94-106 + random.hpp:316-333 restated) is drawn from with engine
steps of minstd_rand0 and held to them by a chi-squared test, and its
likelihoods to the probabilities within fmath::exp's stated error; the -m gpu
test does the same through the C ABI (dist_sample_from_scores_overwrite).
Bit-level parity of this function is unpinned (oracle.h); this pins WHAT IT
SAMPLES to the reference."""
import gzip
import json
import os

import numpy as np
import pytest
from scipy import stats

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(gzip.open(os.path.join(HERE, "golden",
                                         "sampler_probs.json.gz"), "rt"))["cases"]


def chi2_ok(draws, probs, n):
    """pool cells of expectation < 8 into one; p-value of the fit"""
    probs = np.asarray(probs)
    counts = np.bincount(draws, minlength=len(probs)).astype(np.float64)
    big = probs * n >= 8.0
    obs = np.append(counts[big], counts[~big].sum())
    exp = np.append(probs[big] * n, probs[~big].sum() * n)
    keep = exp > 0
    if keep.sum() < 2:
        return 1.0
    return stats.chisquare(obs[keep], exp[keep] * obs[keep].sum()
                           / exp[keep].sum()).pvalue


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_sampler_draws_from_the_reference_softmax(case):
    L = ol.oracle()
    scores = np.asarray(case["scores"], np.float32)
    probs = np.asarray(case["probs"])
    # likelihoods / total == the reference's probabilities, to the first-order
    # table exponential's error (fmath.hpp:438-459: relative 2.3e-7 per entry;
    # the in-order float sum adds K * 6e-8)
    lik = scores.copy()
    st = np.zeros(1, np.uint32)
    st[0] = L.orc_rng_seed(1)
    L.orc_sample_from_scores_overwrite(st.ctypes.data_as(
        ol.ctypes.POINTER(ol.ctypes.c_uint32)), len(lik), lik)
    got = lik.astype(np.float64) / lik.astype(np.float64).sum()
    np.testing.assert_allclose(got, probs, rtol=5e-6, atol=1e-12)
    n = 40000 if len(scores) <= 64 else 150000
    draws = np.empty(n, np.int64)
    for i in range(n):
        buf = scores.copy()
        draws[i] = L.orc_sample_from_scores_overwrite(
            st.ctypes.data_as(ol.ctypes.POINTER(ol.ctypes.c_uint32)),
            len(buf), buf)
    assert chi2_ok(draws, probs, n) > 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if len(c["scores"]) <= 64],
                         ids=[c["name"] for c in CASES
                              if len(c["scores"]) <= 64])
def test_gpu_sampler_draws_from_the_reference_softmax(case):
    from distributions_amd import _core
    scores = np.asarray(case["scores"], np.float32)
    probs = np.asarray(case["probs"])
    st = _core.rng_seed(1)
    n = 20000
    draws = np.empty(n, np.int64)
    for i in range(n):
        buf = scores.copy()
        draws[i], st = _core.sample_from_scores_overwrite(st, buf)
    got = buf.astype(np.float64) / buf.astype(np.float64).sum()
    np.testing.assert_allclose(got, probs, rtol=5e-6, atol=1e-12)
    assert chi2_ok(draws, probs, n) > 1e-4


@pytest.mark.gpu
def test_gpu_sampler_equals_oracle_draw_for_draw():
    """... and the two samplers agree draw for draw (what the bit-exact
    suites rest on), so the distribution-level pin of one is the other's"""
    from distributions_amd import _core
    L = ol.oracle()
    case = next(c for c in CASES if c["name"].startswith("three heavy"))
    scores = np.asarray(case["scores"], np.float32)
    st_g = _core.rng_seed(7)
    st_o = np.array([L.orc_rng_seed(7)], np.uint32)
    for _ in range(300):
        a = scores.copy()
        b = scores.copy()
        g, st_g = _core.sample_from_scores_overwrite(st_g, a)
        o = L.orc_sample_from_scores_overwrite(
            st_o.ctypes.data_as(ol.ctypes.POINTER(ol.ctypes.c_uint32)),
            len(b), b)
        assert g == o and st_g == int(st_o[0])
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
