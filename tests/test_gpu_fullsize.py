"""Parity at BASELINE.json's full sizes (N = 10M rows, K = 1024).

The oracle does ~0.1 M row-updates/s at K = 1024, so it follows the GPU for
ONE sub-sweep of the full-size state bit for bit (every assignment, every
statistic), and the rest of the sweep is checked through size-independent
properties of the domain: the statistics are exactly the recount of the
final assignments ("checksum of checksums"), group sizes sum to N, every
assignment names a live group, and the run is reproducible."""
import numpy as np
import pytest

import oracle_lib as ol
import workloads
from test_gpu_sweep import assert_same_state

pytestmark = pytest.mark.gpu

N = 10_000_000
K = 1024
ALPHA, D = 1.0, 0.2          # bench.py's PitmanYor


def recount_check(gpu, vals, config_kinds):
    """statistics == recount of the assignments"""
    assign = gpu.assignments()
    counts = gpu.counts()
    n_groups = len(gpu)
    assert counts.sum() == N
    assert counts.shape[0] == n_groups
    # global ids -> packed slots through the group sizes: every id in use
    # must be one of the live groups
    ids, sizes = np.unique(assign, return_counts=True)
    assert ids.size == np.count_nonzero(counts)
    assert sorted(sizes.tolist()) == sorted(counts[counts > 0].tolist())
    # statistics per group: find each slot's id through a member row
    # (slot order is the engine's; sizes identify nothing when equal), so
    # compare multisets of (size, statistics) instead
    for f, kind in enumerate(config_kinds):
        v = vals[f]
        if kind == "cat":
            dim = int(v.max()) + 1
            dense = np.zeros((ids.size, dim), np.int64)
            slot = np.searchsorted(ids, assign)
            np.add.at(dense, (slot, v.astype(np.int64)), 1)
            want = sorted(map(tuple, dense.tolist()))
            got = []
            for g in range(n_groups):
                w = gpu.get_group(f, g).astype(np.int64)
                if w[0]:
                    assert w[0] == w[1:].sum()
                    got.append(tuple(w[1:1 + dim].tolist()))
            assert sorted(got) == want
        elif kind == "count":           # GammaPoisson: count, sum
            slot = np.searchsorted(ids, assign)
            sums = np.bincount(slot, weights=v.astype(np.float64),
                               minlength=ids.size).astype(np.int64)
            want = sorted(zip(sizes.tolist(), sums.tolist()))
            got = []
            for g in range(n_groups):
                w = gpu.get_group(f, g)
                if w[0]:
                    got.append((int(w[0]), int(w[1])))
            assert sorted(got) == want
        else:                           # NormalInverseChiSq: count, mean
            slot = np.searchsorted(ids, assign)
            x = v.view(np.float32).astype(np.float64)
            sums = np.bincount(slot, weights=x, minlength=ids.size)
            want = sorted(zip(sizes.tolist(), (sums / sizes).tolist()))
            got = []
            for g in range(n_groups):
                w = gpu.get_group(f, g)
                if w[0]:
                    got.append((int(w[0]),
                                float(w[1:2].view(np.float32)[0])))
            got.sort()
            assert [a for a, _ in got] == [a for a, _ in want]
            # Welford means in f32 over ~10^4 members: 1e-4 absolute
            np.testing.assert_allclose([m for _, m in got],
                                       [m for _, m in want], atol=1e-4)


@pytest.mark.parametrize("config,dim,first,batch,kinds", [
    ("dd", 256, 1_000_000, 1_000_000, ["cat"]),            # BASELINE configs[1]
    ("gp_nich", None, 200_000, 1_000_000, ["count", "real"]),  # configs[2]
])
def test_full_size_sweep(config, dim, first, batch, kinds):
    from distributions_amd import engine
    osh, gsh, vals, assign = workloads.make(config, N, K, dim=dim)
    orc = ol.OracleMixture(ALPHA, D, osh)
    orc.init_from_assignments(vals, assign, K, 1)
    gpu = engine.Gibbs(ALPHA, D, gsh)
    gpu.load_rows(vals, assign, K, 1)
    seed = 20240601
    st = ol.oracle().orc_rng_seed(seed)

    # one sub-sweep of the full-size state, followed by the oracle
    orc.gibbs_batch(0, first, st, 0)
    gpu.sweep(0, first, first, seed, draw_base=0)
    assert_same_state(orc, gpu, "%s first %d rows" % (config, first))

    # the rest of the sweep on the GPU alone: properties
    gpu.sweep(first, N, batch, seed, draw_base=0)
    words = [orc.values[f] for f in range(len(vals))]
    recount_check(gpu, words, kinds)
    final = gpu.assignments().copy()
    moved = np.count_nonzero(final != assign)
    assert moved > N // 2            # the sweep did re-assign rows

    # reproducible: the same seed and batches give the same chain
    again = engine.Gibbs(ALPHA, D, gsh)
    again.load_rows(vals, assign, K, 1)
    again.sweep(0, first, first, seed, draw_base=0)
    again.sweep(first, N, batch, seed, draw_base=0)
    np.testing.assert_array_equal(again.assignments(), final)
    np.testing.assert_array_equal(again.counts(), gpu.counts())


def test_c5_shape_dpd_8192_groups_streamed_tables():
    """BASELINE configs[4]'s shape -- DirichletProcessDiscrete over V = 10 000
    values with K = 8192 groups, likelihood tables (V*K*4 B = 328 MB) streamed
    from HBM by the vector-load form of the value-sorted kernel -- on as many
    rows as the oracle follows in seconds: bit-exact."""
    from distributions_amd import engine
    n, k, dim = 120_000, 8192, 10_000
    osh, gsh, vals, assign = workloads.make("dpd", n, k, dim=dim)
    orc = ol.OracleMixture(0.5, 0.1, osh)
    orc.init_from_assignments(vals, assign, k, 1)
    gpu = engine.Gibbs(0.5, 0.1, gsh)
    gpu.set_option("value_sorted", 2)
    gpu.load_rows(vals, assign, k, 1)
    seed = 77
    st = ol.oracle().orc_rng_seed(seed)
    for b in range(0, n, 60_000):
        orc.gibbs_batch(b, b + 60_000, st, 0)
    gpu.sweep(0, n, 60_000, seed, draw_base=0)
    assert tuple(gpu.path_counts()) == (2, 0)   # value-sorted, generic
    assert_same_state(orc, gpu, "dpd V=10000 K=8192")


def test_more_groups_than_the_lds_paths_hold():
    """K = 16 000 groups: beyond the LDS-aggregated apply kernel (K*4 B > 60
    KiB), the wave-per-row strip and the device-resident chain; the
    direct-atomics, lane-per-row and batch-of-one paths take over"""
    from distributions_amd import engine
    n, k, dim = 48_000, 16_000, 4
    osh, gsh, vals, assign = workloads.make("dd", n, k, dim=dim)
    orc = ol.OracleMixture(1.0, 0.2, osh)
    orc.init_from_assignments(vals, assign, k, 1)
    gpu = engine.Gibbs(1.0, 0.2, gsh)
    gpu.set_option("value_sorted", 2)
    gpu.load_rows(vals, assign, k, 1)
    seed = 5
    st = ol.oracle().orc_rng_seed(seed)
    for b in range(0, n, 16_000):
        orc.gibbs_batch(b, b + 16_000, st, 0)
    gpu.sweep(0, n, 16_000, seed, draw_base=0)
    assert_same_state(orc, gpu, "K=16000 batches")
    state = orc.gibbs_sequential(0, 40, st)
    assert gpu.sweep_sequential(0, 40, st) == state
    assert_same_state(orc, gpu, "K=16000 sequential")
