"""Scan sampling (option "sampling" = 1): the tolerance-level, opt-in mode of
the batched row update -- one score evaluation per (row, group), a running
log-sum-exp, the draw located by cumulative sums.  Same scores (bit for bit)
and the same draw per row as the exact mode; the sampled index follows the
same distribution (random.hpp:316-333, random.cc:94-106) but is computed with
a different float summation order, so it may differ from the exact mode's
where u * total falls within rounding of a boundary between two groups.

(i)   samples against the softmax of the scores, Pearson chi-squared as the
      reference's own check (distributions/tests/test_random.py:183-210,
      tests/util.py:182-203: goodness of fit > 1e-3);
(ii)  agreement with the exact mode on the same batch (> 99.5 % asserted;
      the measured rate is printed);
(iii) the scores are the exact mode's, bit for bit."""
import numpy as np
import pytest
from scipy import stats

import workloads

pytestmark = pytest.mark.gpu

CONFIGS = ["gp_nich", "nich", "nich2", "dd_bb_gp", "gp", "dd", "bnb"]


def engine_for(config, n, k, sampling, fold=1, seed=workloads.SEED,
               extra_rows=None, value_sorted=0):
    from distributions_amd import engine
    osh, gsh, vals, assign = workloads.make(config, n, k, seed=seed)
    if extra_rows is not None:
        count, values, group = extra_rows
        vals = [np.concatenate([v, np.full(count, x, v.dtype)])
                for v, x in zip(vals, values)]
        assign = np.concatenate([assign, np.full(count, group, np.uint32)])
    gpu = engine.Gibbs(1.0, 0.2, gsh)
    gpu.set_option("value_sorted", value_sorted)
    gpu.set_option("sampling", sampling)
    gpu.set_option("debug.rows_fold", fold)
    gpu.load_rows(vals, assign, k, 1)
    return gpu, vals, assign


@pytest.mark.parametrize("config", CONFIGS)
@pytest.mark.parametrize("k", [24, 101])
def test_scan_agrees_with_exact_on_one_batch(config, k):
    n = 50000
    out = []
    for sampling in (0, 1):
        gpu, _, _ = engine_for(config, n, k, sampling)
        scores = gpu.row_scores(123)
        gpu.sweep(0, n, n, 777)          # ONE batch: one common snapshot
        out.append((gpu.assignments().copy(), scores))
        assert gpu.core.debug_counts()["scratch_batches"] == 1
    (exact, s0), (scan, s1) = out
    assert np.array_equal(s0.view(np.uint32), s1.view(np.uint32))
    agree = float((exact == scan).mean())
    first = np.nonzero(exact != scan)[0]
    print("%s K=%d: scan == exact on %.4f %% of %d rows; first divergence %s"
          % (config, k, 100 * agree, n, first[:1]))
    assert agree > 0.995


@pytest.mark.parametrize("config,probe", [("gp_nich", (3, 0.25)),
                                          ("dd_bb_gp", (2, 1, 4)),
                                          ("nich", (-0.5,))])
@pytest.mark.parametrize("fold", [0, 2])
def test_scan_samples_match_scores(config, probe, fold):
    """20 000 identical probe rows in one group: in batch semantics they all
    see the same score vector, so their new groups are draws from its
    softmax."""
    n, k, m = 30000, 12, 20000
    gpu, vals, _ = engine_for(config, n, k, 1, fold=fold,
                              extra_rows=(m, probe, 3))
    scores = gpu.row_scores(n)           # the first probe row
    gpu.sweep(0, n + m, n + m, 4321)
    new = gpu.assignments()[n:]
    # (global ids == packed ids here: one batch, no group created before it)
    counts = np.bincount(new, minlength=len(scores))[:len(scores)]
    assert counts.sum() == m
    p = np.exp(scores.astype(np.float64) - scores.max())
    p /= p.sum()
    keep = p * m >= 5                    # chi-squared needs expected counts
    obs = np.append(counts[keep], counts[~keep].sum())
    exp = np.append(p[keep] * m, p[~keep].sum() * m)
    if exp[-1] == 0:
        obs, exp = obs[:-1], exp[:-1]
    gof = stats.chisquare(obs, exp * obs.sum() / exp.sum()).pvalue
    print("goodness of fit", gof)
    assert gof > 1e-3


VS_CONFIGS = ["dd", "dd_skew", "bb", "gp", "dpd", "dpd_other", "bnb"]


@pytest.mark.parametrize("config", VS_CONFIGS)
@pytest.mark.parametrize("k", [24, 101, 1000])
def test_value_sorted_scan_agrees_with_exact_on_one_batch(config, k):
    """the value-sorted path's scan sampling (per-value prefix sums, a binary
    search per row) against its exact kernels"""
    n = 60000
    out = []
    for sampling in (0, 1):
        gpu, _, _ = engine_for(config, n, k, sampling, value_sorted=2)
        gpu.sweep(0, n, n, 777)
        out.append(gpu.assignments().copy())
        assert gpu.core.debug_counts()["scan_batches"] == sampling
    exact, scan = out
    agree = float((exact == scan).mean())
    first = np.nonzero(exact != scan)[0]
    print("%s K=%d: value-sorted scan == exact on %.4f %% of %d rows; first "
          "divergence %s" % (config, k, 100 * agree, n, first[:1]))
    assert agree > 0.995


@pytest.mark.parametrize("config,probe", [("dd", (5,)), ("gp", (7,)),
                                          ("bb", (1,))])
def test_value_sorted_scan_samples_match_scores(config, probe):
    n, k, m = 30000, 12, 20000
    gpu, _, _ = engine_for(config, n, k, 1, extra_rows=(m, probe, 3),
                           value_sorted=2)
    scores = gpu.row_scores(n)
    gpu.sweep(0, n + m, n + m, 4321)
    assert gpu.core.debug_counts()["scan_batches"] == 1
    new = gpu.assignments()[n:]
    counts = np.bincount(new, minlength=len(scores))[:len(scores)]
    assert counts.sum() == m
    p = np.exp(scores.astype(np.float64) - scores.max())
    p /= p.sum()
    keep = p * m >= 5
    obs = np.append(counts[keep], counts[~keep].sum())
    exp = np.append(p[keep] * m, p[~keep].sum() * m)
    if exp[-1] == 0:
        obs, exp = obs[:-1], exp[:-1]
    gof = stats.chisquare(obs, exp * obs.sum() / exp.sum()).pvalue
    print("goodness of fit", gof)
    assert gof > 1e-3


def test_value_sorted_scan_many_sweeps_keep_the_state_consistent():
    """statistics stay those of the assignments (the apply kernels are the
    exact path's), groups are created and removed, through device-normalised
    runs"""
    n, k = 40000, 200
    gpu, vals, _ = engine_for("dd", n, k, 1, value_sorted=2)
    for sweep in range(6):
        gpu.sweep(0, n, 8000, 99, draw_base=sweep * n)
    assign = gpu.assignments()
    counts = gpu.counts()
    packed = np.array([gpu.core.global_to_packed(int(a)) for a in
                       np.unique(assign)])
    sizes = np.bincount([gpu.core.global_to_packed(int(a)) for a in assign],
                        minlength=len(counts))
    assert np.array_equal(sizes, counts)
    assert packed.max() < len(counts)
    # the categorical counts of a group are those of its rows
    g0 = gpu.core.global_to_packed(int(assign[0]))
    members = np.array([gpu.core.global_to_packed(int(a)) for a in assign]) == g0
    words = gpu.get_group(0, g0)
    assert words[0] == members.sum()
    assert np.array_equal(words[1:17],
                          np.bincount(vals[0][members], minlength=16))


# ---------------------------------------------------------------------------
# "float_stats" = 1: the order-dependent statistics as binary64 sums


def group_stats(gpu, f, kind):
    out = []
    for g in range(len(gpu)):
        w = gpu.get_group(f, g)
        if kind == "nich":
            out.append((int(np.int32(w[0])), float(w[1:2].view(np.float32)[0]),
                        float(w[2:3].view(np.float32)[0])))
        else:   # gp: count, sum, log_prod
            out.append((int(w[0]), int(w[1]),
                        float(w[2:3].view(np.float32)[0])))
    return np.array(out, np.float64)


@pytest.mark.parametrize("config,kinds", [("gp_nich", ("gp", "nich")),
                                          ("nich", ("nich",)),
                                          ("gp", ("gp",))])
def test_merged_float_stats_match_the_ordered_replay_to_rounding(config, kinds):
    """one batch from the same state: the same moves, statistics equal to the
    sequential updates' up to binary32 rounding"""
    n, k = 40000, 50
    res = []
    for merged in (0, 1):
        gpu, _, _ = engine_for(config, n, k, 0)
        gpu.set_option("float_stats", merged)
        gpu.sweep(0, n, n, 2024)
        res.append((gpu.assignments().copy(),
                    [group_stats(gpu, f, kind) for f, kind in enumerate(kinds)],
                    gpu.core.debug_counts()["merged_batches"]))
    (a0, s0, m0), (a1, s1, m1) = res
    assert (m0, m1) == (0, 1)
    assert np.array_equal(a0, a1)
    for x, y in zip(s0, s1):
        assert np.array_equal(x[:, 0], y[:, 0])            # counts: exact
        np.testing.assert_allclose(y[:, 1:], x[:, 1:], rtol=2e-5, atol=2e-5)


def test_merged_float_stats_stay_those_of_the_rows_over_sweeps():
    """ten sweeps of sub-sweeps (scan sampling + merged statistics: the
    tolerance-level configuration): every group's count / mean /
    count_times_variance are those of its rows, computed in binary64"""
    n, k = 30000, 20
    gpu, vals, _ = engine_for("gp_nich", n, k, 1)
    gpu.set_option("float_stats", 1)
    for sweep in range(10):
        gpu.sweep(0, n, 5000, 11, draw_base=sweep * n)
    assign = gpu.assignments()
    packed = np.array([gpu.core.global_to_packed(int(a)) for a in assign])
    got = group_stats(gpu, 1, "nich")
    x = vals[1].astype(np.float64)
    for g in range(len(gpu)):
        rows = x[packed == g]
        assert got[g, 0] == len(rows)
        if len(rows):
            assert abs(got[g, 1] - rows.mean()) < 1e-4 * max(1, abs(rows.mean()))
        if len(rows) > 1:
            ctv = ((rows - rows.mean()) ** 2).sum()
            assert abs(got[g, 2] - ctv) < 1e-3 * max(1.0, ctv)
    lf = np.array([float(np.sum([np.log(np.arange(1, v + 1)).sum()
                                 for v in vals[0][packed == g]]))
                   for g in range(len(gpu))])
    got_gp = group_stats(gpu, 0, "gp")
    np.testing.assert_allclose(got_gp[:, 2], lf, rtol=1e-3, atol=1e-2)
