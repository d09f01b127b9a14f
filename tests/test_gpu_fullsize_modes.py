"""The opt-in, tolerance-level modes at BASELINE.json's full sizes: scan
sampling (option "sampling" = 1) on C2, C3 and C5 at N = 10M, merged float
statistics ("float_stats" = 1) on C3's full-size state.  What the modes
promise is checked where bench.py quotes their numbers:

* the scores are the exact mode's, bit for bit;
* on one sub-sweep from a common state the sampled groups agree with the
  exact mode's on > 99.5 % of the rows at K = 1024 (the draws differ only
  where u * total falls within rounding of a boundary) -- and on > 96 % at
  K = 8192 (measured: 97.5 %): there the reference's own in-order float sum
  carries a rounding error of the order of a group's share of the total
  (K/2 ulps against 1/K), which no other summation order reproduces; both are
  draws from the softmax to float accuracy, which the chi-squared test below
  checks at that K;
* after a whole sweep the statistics are the recount of the assignments;
* draws follow the softmax of the scores at K = 1024 and K = 8192 (Pearson
  chi-squared as distributions/tests/test_random.py:183-210);
* merged float statistics equal the ordered replay's to binary32 rounding
  after one sub-sweep (same assignments, same counts), and the rows' own
  moments after a sweep."""
import numpy as np
import pytest
from scipy import stats

import workloads
from test_gpu_fullsize import ALPHA, D, K, N, recount_check

pytestmark = pytest.mark.gpu


def pair(config, k, dim, options):
    """two engines on the same rows: the exact mode and `options`"""
    from distributions_amd import engine
    osh, gsh, vals, assign = workloads.make(config, N, k, dim=dim)
    out = []
    for opts in ({}, options):
        gpu = engine.Gibbs(ALPHA, D, gsh)
        for name, value in opts.items():
            gpu.set_option(name, value)
        gpu.load_rows(vals, assign, k, 1)
        out.append(gpu)
    return out[0], out[1], vals, assign


@pytest.mark.parametrize("config,k,dim,kinds,probes,floor", [
    ("dd", K, 256, ["cat"], 6, 0.995),                  # BASELINE configs[1]
    ("gp_nich", K, None, ["count", "real"], 6, 0.995),  # configs[2]
    ("dpd", 8192, 10_000, ["cat"], 2, 0.96),            # configs[4]
])
def test_scan_mode_at_full_size(config, k, dim, kinds, probes, floor):
    exact, scan, vals, assign = pair(config, k, dim, {"sampling": 1})
    batch = 1_000_000
    rows = np.linspace(0, batch - 1, probes).astype(int)
    for r in rows:      # (i) the scores are the exact mode's
        a, b = exact.row_scores(int(r)), scan.row_scores(int(r))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    seed = 20240601
    exact.sweep(0, batch, batch, seed, draw_base=0)
    scan.sweep(0, batch, batch, seed, draw_base=0)
    e, s = exact.assignments()[:batch], scan.assignments()[:batch]
    agree = float((e == s).mean())
    print("%s: scan == exact on %.4f %% of %d rows (K = %d)"
          % (config, 100 * agree, batch, k))
    assert agree > floor                      # (ii)
    assert np.count_nonzero(s != assign[:batch]) > batch // 2
    del exact
    # (iii) the rest of the sweep in scan mode: statistics == recount
    scan.sweep(batch, N, batch, seed, draw_base=0)
    counts = scan.core.debug_counts()
    assert counts["scan_batches"] == (10 if config != "gp_nich" else 0)
    recount_check(scan, vals, kinds)


@pytest.mark.parametrize("config,value_sorted,k,probe", [
    ("dd", 2, 1024, (3,)), ("dd", 2, 8192, (3,)),
    ("gp_nich", 0, 1024, (3, 0.25)), ("dpd", 2, 8192, (3,)),
])
def test_scan_draws_follow_the_softmax_at_full_group_counts(
        config, value_sorted, k, probe):
    """Identical probe rows in one group of a K-group mixture: in batch
    semantics they see one score vector, their new groups are draws from its
    softmax (test_gpu_scan.py's check at K = 12, here at the group counts of
    BASELINE configs[1] and [4]).  A thousand rows per group keep the vector
    flat enough for (nearly) every group to expect at least five draws: the
    chi-squared runs over about K cells."""
    from distributions_amd import engine
    n, m = 1000 * k, (20000 if k <= 1024 else 100000)
    osh, gsh, vals, assign = workloads.make(config, n, k)
    vals = [np.concatenate([v, np.full(m, x, v.dtype)])
            for v, x in zip(vals, probe)]
    assign = np.concatenate([assign, np.full(m, 3, np.uint32)])
    gpu = engine.Gibbs(ALPHA, D, gsh)
    gpu.set_option("value_sorted", value_sorted)
    gpu.set_option("sampling", 1)
    gpu.load_rows(vals, assign, k, 1)
    scores = gpu.row_scores(n)
    gpu.sweep(0, n + m, n + m, 4321)
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
    print("%s K=%d goodness of fit %.3g over %d cells" % (config, k, gof,
                                                          obs.size))
    assert obs.size > k // 2
    assert gof > 1e-3


def test_merged_float_statistics_on_c3_at_full_size():
    """float_stats = 1 (NICH count / mean / ctv and GP log_prod from binary64
    sums per group) against the ordered replay, C3 at N = 10M, K = 1024."""
    ordered, merged, vals, assign = pair("gp_nich", K, None, {"float_stats": 1})
    batch = 1_000_000
    seed = 20240601
    ordered.sweep(0, batch, batch, seed, draw_base=0)
    merged.sweep(0, batch, batch, seed, draw_base=0)
    assert merged.core.debug_counts()["merged_batches"] == 1
    # one sub-sweep from a common state: the same moves ...
    np.testing.assert_array_equal(ordered.assignments(), merged.assignments())
    np.testing.assert_array_equal(ordered.counts(), merged.counts())
    # ... and float statistics equal to binary32 rounding
    worst = 0.0
    for g in range(0, len(ordered), 7):
        for f in (0, 1):
            a = ordered.get_group(f, g).view(np.float32).astype(np.float64)
            b = merged.get_group(f, g).view(np.float32).astype(np.float64)
            ia, ib = ordered.get_group(f, g), merged.get_group(f, g)
            if f == 0:     # GP: count, sum (integers), log_prod (float)
                assert ia[0] == ib[0] and ia[1] == ib[1]
                fa, fb = a[2:3], b[2:3]
            else:          # NICH: count (integer), mean, ctv
                assert ia[0] == ib[0]
                fa, fb = a[1:3], b[1:3]
            worst = max(worst, float(np.max(
                np.abs(fa - fb) / (1e-3 + np.abs(fa)))))
    print("merged vs ordered float statistics after one sub-sweep: worst "
          "relative difference %.3g" % worst)
    assert worst < 2e-5
    del ordered
    merged.sweep(batch, N, batch, seed, draw_base=0)
    recount_check(merged, vals, ["count", "real"])
