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


@pytest.mark.parametrize("config,dim,first,batch,kinds,d", [
    ("dd", 256, 1_000_000, 1_000_000, ["cat"], D),         # BASELINE configs[1]
    ("gp_nich", None, 200_000, 1_000_000, ["count", "real"], D),  # configs[2]
    # SURVEY 8d's variants of C2: Zipf(1.1) values; d = 0 (the CRP)
    ("dd_zipf", 256, 300_000, 1_000_000, ["cat"], D),
    ("dd", 256, 300_000, 1_000_000, ["cat"], 0.0),
])
def test_full_size_sweep(config, dim, first, batch, kinds, d):
    from distributions_amd import engine
    D = d
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


@pytest.mark.parametrize("batch", [1_000_000, 65_536])
def test_steady_state_sub_sweep_at_full_size(batch):
    """The state bench.py times: after a whole sweep at N = 10M / K = 1024 the
    tiles of every batch range are sorted by group, each value's arg-max rows
    have a tile of their own and totals start from the per-value running
    sums.  The oracle adopts the GPU's state after sweep 1 (group order,
    statistics, ids, assignments) and follows the first sub-sweep of sweep 2
    bit for bit."""
    from distributions_amd import engine
    osh, gsh, vals, assign = workloads.make("dd", N, K, dim=256)
    gpu = engine.Gibbs(ALPHA, D, gsh)
    gpu.load_rows(vals, assign, K, 1)
    seed = 20240601
    st = ol.oracle().orc_rng_seed(seed)
    gpu.sweep(0, N, batch, seed, draw_base=0)
    assert gpu.counts().sum() == N
    orc = ol.OracleMixture(ALPHA, D, osh)
    orc.adopt(gpu, vals)
    assert_same_state(orc, gpu, "adopted state")
    before = gpu.core.debug_counts()

    orc.gibbs_batch(0, batch, st, N)
    gpu.sweep(0, batch, batch, seed, draw_base=N)
    after = gpu.core.debug_counts()
    assert after["value_sorted_batches"] == before["value_sorted_batches"] + 1
    assert after["other_batches"] == before["other_batches"] == 0
    if batch >= 1_000_000:
        # the launch is large enough for both devices, and the group-sorted
        # range lets (nearly) every value give its arg-max rows their tile
        assert after["band_launches"] == before["band_launches"] + 1
        assert after["running_sum_launches"] == before["running_sum_launches"] + 1
        assert after["band_values_last"] >= 200
    assert_same_state(orc, gpu, "sweep 2, first sub-sweep of %d" % batch)

    # and the second sub-sweep of sweep 2 (the caches of a range sampled a
    # moment ago next to one that was not)
    orc.gibbs_batch(batch, 2 * batch, st, N)
    gpu.sweep(batch, 2 * batch, batch, seed, draw_base=N)
    assert_same_state(orc, gpu, "sweep 2, second sub-sweep of %d" % batch)


def test_c3_steady_state_sub_sweep_at_full_size():
    """BASELINE configs[2] past its first sweep: GammaPoisson +
    NormalInverseChiSq rows, N = 10M, K = 1024.  After a whole sweep on the
    GPU (ordered replay of the Welford statistics and the log-products
    included) the oracle adopts the engine's state -- float statistics bit
    for bit -- and follows a sub-sweep of sweep 2."""
    from distributions_amd import engine
    osh, gsh, vals, assign = workloads.make("gp_nich", N, K)
    gpu = engine.Gibbs(ALPHA, D, gsh)
    gpu.load_rows(vals, assign, K, 1)
    seed = 20240601
    st = ol.oracle().orc_rng_seed(seed)
    gpu.sweep(0, N, 1_000_000, seed, draw_base=0)
    assert gpu.counts().sum() == N
    orc = ol.OracleMixture(ALPHA, D, osh)
    orc.adopt(gpu, vals)
    assert_same_state(orc, gpu, "adopted state")
    first = 150_000
    orc.gibbs_batch(0, first, st, N)
    gpu.sweep(0, first, first, seed, draw_base=N)
    assert_same_state(orc, gpu, "C3 sweep 2, first %d rows" % first)


def test_c1_dd16_k64_n100k():
    """BASELINE configs[0] (benchmarks/mixture.cc's model at its CPU size):
    DirichletDiscrete(dim=16), K = 64, N = 100 000 -- frozen sub-sweeps of
    4096 / 65 536 / N rows and then the reference's sequential chain over the
    whole table, each followed by the oracle bit for bit."""
    from distributions_amd import engine
    n, k, dim = 100_000, 64, 16
    osh, gsh, vals, assign = workloads.make("dd", n, k, dim=dim)
    orc = ol.OracleMixture(ALPHA, 0.0, osh)       # CRP for C1/C2
    orc.init_from_assignments(vals, assign, k, 1)
    gpu = engine.Gibbs(ALPHA, 0.0, gsh)
    gpu.load_rows(vals, assign, k, 1)
    seed = 20240601
    st = ol.oracle().orc_rng_seed(seed)
    sweep = 0
    for batch in (4096, 65_536, n):
        for b in range(0, n, batch):
            orc.gibbs_batch(b, min(n, b + batch), st, sweep * n)
        gpu.sweep(0, n, batch, seed, draw_base=sweep * n)
        assert_same_state(orc, gpu, "C1 batch %d" % batch)
        sweep += 1
    state = orc.gibbs_sequential(0, n, st)
    assert gpu.sweep_sequential(0, n, st) == state
    assert_same_state(orc, gpu, "C1 sequential sweep")


def test_c5_full_size_dpd_10m_rows():
    """BASELINE configs[4] at its full size: DirichletProcessDiscrete over
    V = 10 000 values, K = 8192 groups, N = 10M rows, bench.py's
    PitmanYor(1, 0.2).  The oracle (12 k row-updates/s at this K) follows one
    sub-sweep of 20 000 rows bit for bit; the rest of the sweep runs in
    sub-sweeps of 10^6 rows and is checked through the recount properties."""
    from distributions_amd import engine
    k, dim, first = 8192, 10_000, 20_000
    osh, gsh, vals, assign = workloads.make("dpd", N, k, dim=dim)
    orc = ol.OracleMixture(ALPHA, D, osh)
    orc.init_from_assignments(vals, assign, k, 1)
    gpu = engine.Gibbs(ALPHA, D, gsh)
    gpu.load_rows(vals, assign, k, 1)
    seed = 99
    st = ol.oracle().orc_rng_seed(seed)
    orc.gibbs_batch(0, first, st, 0)
    gpu.sweep(0, first, first, seed, draw_base=0)
    assert len(gpu) == len(orc)
    np.testing.assert_array_equal(gpu.counts(), orc.counts())
    np.testing.assert_array_equal(gpu.assignments(), orc.assign)
    for g in range(0, len(orc), 97):      # (a group is 10 001 words)
        np.testing.assert_array_equal(gpu.get_group(0, g),
                                      orc.get_group(0, g))
    del orc
    gpu.sweep(first, N, 1_000_000, seed, draw_base=0)
    # (the 20 000-row batch is below the value-sorted kernel's threshold of
    # 16 rows per value; the ten batches of the sweep proper are above it,
    # and with 100 rows per value they take the table-free kernel)
    assert tuple(gpu.path_counts()) == (10, 1)
    assert gpu.core.debug_counts()["stream_batches"] == 10
    final = gpu.assignments()
    counts = gpu.counts()
    assert counts.sum() == N and counts.shape[0] == len(gpu)
    ids, sizes = np.unique(final, return_counts=True)
    assert ids.size == np.count_nonzero(counts)
    assert sorted(sizes.tolist()) == sorted(counts[counts > 0].tolist())
    # per group: its statistics are the recount of its rows' values
    slot_of = {int(gpu.core.packed_to_global(s)): s for s in range(len(gpu))}
    for gid in ids[::257]:
        w = gpu.get_group(0, slot_of[int(gid)]).astype(np.int64)
        want = np.bincount(vals[0][final == gid], minlength=dim)
        assert w[0] == want.sum()
        np.testing.assert_array_equal(w[1:1 + dim], want)
    assert np.count_nonzero(final != assign) > N // 2


@pytest.mark.parametrize("stream", [0, 2])
def test_c5_shape_dpd_8192_groups_streamed_tables(stream):
    """BASELINE configs[4]'s shape -- DirichletProcessDiscrete over V = 10 000
    values with K = 8192 groups -- on as many rows as the oracle follows in
    seconds, once through the per-value tables (V*K*4 B = 328 MB each,
    streamed from HBM) and once through the table-free kernel: bit-exact."""
    from distributions_amd import engine
    n, k, dim = 120_000, 8192, 10_000
    osh, gsh, vals, assign = workloads.make("dpd", n, k, dim=dim)
    orc = ol.OracleMixture(0.5, 0.1, osh)
    orc.init_from_assignments(vals, assign, k, 1)
    gpu = engine.Gibbs(0.5, 0.1, gsh)
    gpu.set_option("value_sorted", 2)
    gpu.set_option("value_stream", stream)
    gpu.load_rows(vals, assign, k, 1)
    seed = 77
    st = ol.oracle().orc_rng_seed(seed)
    for b in range(0, n, 60_000):
        orc.gibbs_batch(b, b + 60_000, st, 0)
    gpu.sweep(0, n, 60_000, seed, draw_base=0)
    assert tuple(gpu.path_counts()) == (2, 0)   # value-sorted, generic
    assert gpu.core.debug_counts()["stream_batches"] == stream
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


def test_c5_stream_kernel_where_the_bench_runs_it():
    """k_vs_stream in the shape bench.py times for BASELINE configs[4]:
    K = 8192 groups, a Shared of V = 10 000 values, ~100 rows per value and
    sub-sweep, tiles re-sorted by group (sweep >= 2).  200 000 rows over
    2 000 of the values; sweep 1 on the GPU (one sub-sweep, table-free
    kernel), the oracle adopts the state and follows ALL of sweep 2 -- the
    stream launch itself, not a generic-kernel stand-in -- bit for bit
    (dpd.hpp:517-543, mixture.hpp:84-119)."""
    from distributions_amd import engine
    n, k, dim = 200_000, 8192, 10_000
    osh, gsh, vals, assign = workloads.make("dpd", n, k, dim=dim)
    rs = np.random.default_rng(8)
    # (2 000 of the 10 000 values: 20 rows per value of the domain keep the
    # sub-sweep on the value-sorted path, 100 per value in use make it one
    # tile each -- the table-free kernel's case, chosen by the library itself)
    vals = [(rs.integers(0, 2000, n) * 5 + 3).astype(np.uint32)]
    gpu = engine.Gibbs(ALPHA, D, gsh)
    gpu.load_rows(vals, assign, k, 1)
    seed = 31337
    st = ol.oracle().orc_rng_seed(seed)
    gpu.sweep(0, n, n, seed, draw_base=0)
    before = gpu.core.debug_counts()
    assert before["stream_batches"] == 1 and before["other_batches"] == 0
    assert gpu.validate()["code"] == 0
    orc = ol.OracleMixture(ALPHA, D, osh)
    orc.adopt(gpu, vals)
    for sweep in (1,):
        orc.gibbs_batch(0, n, st, sweep * n)
        gpu.sweep(0, n, n, seed, draw_base=sweep * n)
        after = gpu.core.debug_counts()
        assert after["stream_batches"] == 1 + sweep
        assert after["other_batches"] == 0
        assert len(gpu) == len(orc)
        np.testing.assert_array_equal(gpu.counts(), orc.counts())
        np.testing.assert_array_equal(gpu.assignments(), orc.assign)
    for g in range(0, len(orc), 61):      # (a group is 10 001 words)
        np.testing.assert_array_equal(gpu.get_group(0, g),
                                      orc.get_group(0, g))
    assert gpu.validate()["code"] == 0
