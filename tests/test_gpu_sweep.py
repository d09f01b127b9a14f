"""GPU parity of the batched row update against the oracle (bit-exact
assignment indices, integer statistics and float statistics; scores to the
tolerance stated in each test)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
import workloads

pytestmark = pytest.mark.gpu


VS_ELIGIBLE = ("dd", "dd_skew", "bb", "dpd", "dpd_other", "gp", "bnb")


def both(config, n, k, alpha, d, empty=1, dim=None, seed=workloads.SEED,
         mode=None):
    """mode: None = library default; 0 = generic kernel only; 2 = the
    value-sorted kernel (per-value tables, tiles of 128 rows) whenever the
    feature list allows it; 3 = its table-free form (k_vs_stream) whenever it
    does; 4, 5 = its small-launch form (k_vs_narrow: tiles of 64 rows, vectors
    in LDS, read a whole / half a chunk ahead) whenever it does."""
    from distributions_amd import engine
    osh, gsh, vals, assign = workloads.make(config, n, k, seed=seed, dim=dim)
    orc = ol.OracleMixture(alpha, d, osh)
    orc.init_from_assignments(vals, assign, k, empty)
    gpu = engine.Gibbs(alpha, d, gsh)
    if mode is not None:
        gpu.set_option("value_sorted", min(mode, 2))
        gpu.set_option("value_stream", 2 if mode == 3 else 0)
        gpu.set_option("narrow_tiles", 2 if mode in (4, 5) else 0)
        gpu.set_option("debug.narrow_read_ahead", {4: 8, 5: 4}.get(mode, 0))
    gpu.load_rows(vals, assign, k, empty)
    return orc, gpu


def assert_same_state(orc, gpu, what=""):
    assert len(gpu) == len(orc), what
    np.testing.assert_array_equal(gpu.counts(), orc.counts(), err_msg=what)
    got, want = gpu.assignments(), orc.assign
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "%s first divergent row %d: gpu %d oracle %d" % (
        what, bad[0], got[bad[0]], want[bad[0]])
    for f in range(orc.F):
        for g in range(len(orc)):
            np.testing.assert_array_equal(
                gpu.get_group(f, g), orc.get_group(f, g),
                err_msg="%s feature %d group %d" % (what, f, g))


CONFIGS = ["dd", "dd_skew", "bb", "gp", "nich", "gp_nich", "dpd",
           "dpd_other", "dd_bb_gp", "bnb"]


@pytest.mark.parametrize("config", CONFIGS)
def test_load_matches_oracle(config):
    orc, gpu = both(config, 3000, 24, 1.0, 0.2)
    assert_same_state(orc, gpu, "after load")


@pytest.mark.parametrize("config", CONFIGS)
def test_row_scores_match_oracle(config):
    """scores[k] of a row in batch semantics; tolerance 0 ulp expected (same
    float operations), asserted as bit equality with a 1e-6 relative fallback
    report."""
    orc, gpu = both(config, 2000, 16, 1.0, 0.1)
    for row in [0, 1, 17, 999, 1999]:
        g = int(ol.oracle().orc_mix_global_to_packed(orc.h, int(orc.assign[row])))
        want = orc.row_scores(row, g)
        got = gpu.row_scores(row)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=0)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (
            config, row, np.abs(got - want).max())


@pytest.mark.parametrize("config", CONFIGS)
@pytest.mark.parametrize("batch", [256, 1000, 4096])
@pytest.mark.parametrize("mode", [0, 2, 3, 4, 5])
def test_batch_sweeps_bit_exact(config, batch, mode):
    if mode >= 2 and config not in VS_ELIGIBLE:
        pytest.skip("value-sorted kernel needs one small-domain feature")
    n, k = 4096, 32
    orc, gpu = both(config, n, k, 1.0, 0.2, mode=mode)
    seed = 12345
    st = ol.oracle().orc_rng_seed(seed)
    for sweep in range(2):
        base = sweep * n
        for b in range(0, n, batch):
            orc.gibbs_batch(b, min(n, b + batch), st, base)
        gpu.sweep(0, n, batch, seed, draw_base=base)
        assert_same_state(orc, gpu, "%s sweep %d batch %d" % (config, sweep, batch))
    vs, generic = gpu.path_counts()
    assert (vs > 0 and generic == 0) if mode >= 2 else (vs == 0)
    counts = gpu.core.debug_counts()
    assert counts["stream_batches"] == (vs if mode == 3 else 0)
    assert counts["narrow_batches"] == (vs if mode in (4, 5) else 0)


@pytest.mark.parametrize("config,dim", [("dd", 4), ("dpd", 6), ("bb", None)])
def test_stream_kernel_on_full_tiles(config, dim):
    """k_vs_stream deals a tile's rows to lanes of one class each, which can
    cost a full tile one slot: the host cuts tiles of 126 rows for it, but a
    range sorted while the tables were in use keeps its tiles of 128 -- the
    row without a lane is handed over.  Few values, so every tile is full."""
    n, k = 4096, 12
    orc, gpu = both(config, n, k, 1.0, 0.2, dim=dim, mode=2)
    st = ol.oracle().orc_rng_seed(99)
    for sweep in range(4):
        if sweep == 1:
            gpu.set_option("value_stream", 2)   # (the range is sorted already)
        base = sweep * n
        orc.gibbs_batch(0, n, st, base)
        gpu.sweep(0, n, n, 99, draw_base=base)
        assert_same_state(orc, gpu, "%s sweep %d" % (config, sweep))
    assert gpu.core.debug_counts()["stream_batches"] == 3


@pytest.mark.parametrize("config", CONFIGS + ["nich2"])
@pytest.mark.parametrize("scratch,lds_log,block,fold",
                         [(0, 1, 512, 0), (3, 0, 256, 2), (3, 1, 1024, 0),
                          (3, 0, 64, 2), (3, 1, 512, 0), (3, 1, 256, 2),
                          (3, 1, 64, 2)])
@pytest.mark.parametrize("k", [31, 40])
def test_general_rows_scratch_kernel_bit_exact(config, scratch, lds_log, block,
                                               fold, k):
    """k_rows_scratch (scratch = 3) and k_sweep_program (0) against the oracle
    on every feature list, with group counts on either side of the blocks of
    8 groups.  Small workgroups give every wave several row tiles; the table
    of FastLog in LDS or not.  fold 2: the leading
    discrete features' scores from the per-(joint value, group) table, rows
    sorted by joint value, whenever the joint domain fits the batch."""
    n = 6000
    from distributions_amd import engine
    osh, gsh, vals, assign = workloads.make(config, n, k)
    orc = ol.OracleMixture(1.0, 0.2, osh)
    orc.init_from_assignments(vals, assign, k, 1)
    gpu = engine.Gibbs(1.0, 0.2, gsh)
    gpu.set_option("value_sorted", 0)
    gpu.set_option("debug.rows_scratch", scratch)
    gpu.set_option("debug.rows_scratch_lds_log", lds_log)
    gpu.set_option("debug.rows_scratch_block", block)
    gpu.set_option("debug.rows_fold", fold)
    gpu.load_rows(vals, assign, k, 1)
    seed = 4242
    st = ol.oracle().orc_rng_seed(seed)
    for sweep in range(2):
        for b in range(0, n, 3000):
            orc.gibbs_batch(b, b + 3000, st, sweep * n)
        gpu.sweep(0, n, 3000, seed, draw_base=sweep * n)
        assert_same_state(orc, gpu, "%s sweep %d" % (config, sweep))
    counts = gpu.core.debug_counts()
    assert counts["scratch_batches"] == (4 if scratch else 0)
    discrete_first = config not in ("nich", "nich2")
    assert counts["fold_batches"] == (
        4 if scratch and fold and discrete_first else 0)


@pytest.mark.parametrize("config,dim,k", [("dd", 256, 64), ("dd_skew", 64, 16),
                                          ("dpd_other", 300, 24), ("bb", None, 8),
                                          ("gp", None, 12)])
@pytest.mark.parametrize("mode", [None, 2, 3, 4, 5])
def test_value_sorted_larger_batches(config, dim, k, mode):
    """default mode picks the value-sorted kernel for large batches; groups of
    very different sizes make rows sit in the arg-max group (class B)."""
    n = 60000
    orc, gpu = both(config, n, k, 1.0, 0.1, dim=dim, mode=mode)
    seed = 2024
    st = ol.oracle().orc_rng_seed(seed)
    for sweep in range(3):
        for b in range(0, n, 20000):
            orc.gibbs_batch(b, b + 20000, st, sweep * n)
        gpu.sweep(0, n, 20000, seed, draw_base=sweep * n)
        assert_same_state(orc, gpu, "%s sweep %d" % (config, sweep))
    assert gpu.path_counts()[0] == 9


@pytest.mark.parametrize("empty", [1, 2])
def test_device_normalised_runs_stay_open_across_sweeps(empty):
    """A device-normalised sweep leaves its run open: the next sweep goes on
    with it and the host's mirrors are pulled by the first call that needs
    them.  Twenty sweeps without a look at the state (the room reserved for
    eight is used up twice), then a different tiling, a sequential stretch
    and more sweeps: the oracle agrees at every look."""
    from distributions_amd import engine
    n, k = 6000, 300
    osh, gsh, vals, assign = workloads.make("dd", n, k, dim=16)
    orc = ol.OracleMixture(20.0, 0.5, osh)
    orc.init_from_assignments(vals, assign, k, empty)
    gpu = engine.Gibbs(20.0, 0.5, gsh)
    gpu.set_option("value_sorted", 2)
    gpu.set_option("device_normalise", 1)
    gpu.load_rows(vals, assign, k, empty)
    seed = 31337
    st = ol.oracle().orc_rng_seed(seed)
    draws = 0

    def sweeps(count, batch):
        nonlocal draws
        for _ in range(count):
            for b in range(0, n, batch):
                orc.gibbs_batch(b, min(n, b + batch), st, draws)
            gpu.sweep(0, n, batch, seed, draw_base=draws)
            draws += n
    sweeps(20, 1500)
    assert_same_state(orc, gpu, "after 20 unobserved sweeps")
    sweeps(3, 1500)
    sweeps(2, 1000)          # another tiling: the open run goes on or is closed
    assert_same_state(orc, gpu, "after a change of tiling")
    sweeps(1, 1500)
    rng = ol.oracle().orc_rng_seed(7)
    assert orc.gibbs_sequential(100, 160, rng) == gpu.sweep_sequential(
        100, 160, rng)                   # (closes the open run first)
    sweeps(2, 1500)
    assert_same_state(orc, gpu, "after a sequential stretch")
    assert gpu.core.debug_counts()["device_normalised"] >= 4 * 28


@pytest.mark.parametrize("normalise", [0, 1])
def test_kernel_timing_samples_every_nth_batch(normalise):
    """kernel_stats counts the batches whose score+sample kernel sat between
    two events: every batch by default, every n-th with kernel_timing = n,
    none with 0 -- and the results do not depend on it"""
    from distributions_amd import engine
    n, k = 8000, 50
    osh, gsh, vals, assign = workloads.make("dd", n, k, dim=16)
    finals = []
    for every, want in ((1, 16), (4, 4), (0, 0)):
        gpu = engine.Gibbs(1.0, 0.2, gsh)
        gpu.set_option("value_sorted", 2)
        gpu.set_option("device_normalise", normalise)
        gpu.set_option("kernel_timing", every)
        gpu.load_rows(vals, assign, k, 1)
        gpu.kernel_stats(reset=True)
        for sweep in range(2):
            gpu.sweep(0, n, 1000, 99, draw_base=sweep * n)
        ms, launches, rows = gpu.kernel_stats()
        assert launches == want and rows == want * 1000
        assert (ms > 0) == (want > 0)
        finals.append(gpu.assignments().copy())
    assert np.array_equal(finals[0], finals[1])
    assert np.array_equal(finals[0], finals[2])


def test_open_run_is_closed_by_whatever_comes_next():
    """options set, rows reloaded, the engine dropped while a device-normalised
    run is open: each finds the state the oracle has"""
    from distributions_amd import engine
    n, k = 5000, 120
    osh, gsh, vals, assign = workloads.make("dd", n, k, dim=16)
    seed = 555
    st = ol.oracle().orc_rng_seed(seed)

    def fresh():
        orc = ol.OracleMixture(5.0, 0.3, osh)
        orc.init_from_assignments(vals, assign, k, 1)
        gpu = engine.Gibbs(5.0, 0.3, gsh)
        gpu.set_option("value_sorted", 2)
        gpu.load_rows(vals, assign, k, 1)
        return orc, gpu
    orc, gpu = fresh()
    for sweep in range(3):
        for b in range(0, n, 1250):
            orc.gibbs_batch(b, b + 1250, st, sweep * n)
        gpu.sweep(0, n, 1250, seed, draw_base=sweep * n)
    gpu.set_option("narrow_tiles", 0)          # (drops the cached ranges)
    for b in range(0, n, 1250):
        orc.gibbs_batch(b, b + 1250, st, 3 * n)
    gpu.sweep(0, n, 1250, seed, draw_base=3 * n)
    assert gpu.core.debug_counts()["device_normalised"] == 16
    assert_same_state(orc, gpu, "after an option change between sweeps")
    gpu.sweep(0, n, 1250, seed, draw_base=4 * n)
    gpu.load_rows(vals, assign, k, 1)          # reload under an open run
    orc2, _ = fresh()
    assert_same_state(orc2, gpu, "after a reload")
    gpu.sweep(0, n, 1250, seed, draw_base=0)
    del gpu                                    # destroyed with the run open
    _core_sync = __import__("distributions_amd._core", fromlist=["x"])
    _core_sync.synchronize()


@pytest.mark.parametrize("config,dim,k", [("dd", 256, 64), ("dd_skew", 64, 16),
                                          ("dpd_other", 300, 24), ("bb", None, 8),
                                          ("gp", None, 12), ("dd", 16, 700)])
def test_value_sorted_running_sums(config, dim, k):
    """the per-value running sums (a tile's total starts at its first own
    chunk) are a tuning choice of large launches: forced on here, same bits"""
    n = 60000
    orc, gpu = both(config, n, k, 1.0, 0.1, dim=dim)
    gpu.set_option("value_sorted", 2)
    gpu.set_option("debug.running_sums_min_tiles", 0)
    seed = 77
    st = ol.oracle().orc_rng_seed(seed)
    for sweep in range(3):
        for b in range(0, n, 30000):
            orc.gibbs_batch(b, b + 30000, st, sweep * n)
        gpu.sweep(0, n, 30000, seed, draw_base=sweep * n)
        assert_same_state(orc, gpu, "%s sweep %d" % (config, sweep))
    assert gpu.path_counts() == (6, 0)


@pytest.mark.parametrize("config", ["dd", "gp_nich", "bb"])
def test_sequential_chain_bit_exact(config):
    """batch of one row == the reference's sequential update."""
    n, k = 300, 8
    orc, gpu = both(config, n, k, 1.0, 0.2)
    st = ol.oracle().orc_rng_seed(7)
    st_o = orc.gibbs_sequential(0, n, st)
    st_g = gpu.sweep_sequential(0, n, st)
    assert st_o == st_g
    assert_same_state(orc, gpu, config + " sequential")


@pytest.mark.parametrize("config,mode", [("dd", 0), ("dd", 2), ("gp_nich", 0),
                                         ("bb", 2), ("gp", 2)])
@pytest.mark.parametrize("empty", [1, 3])
def test_group_creation_and_removal(config, mode, empty):
    """Many groups, few rows, large alpha: rows are alone in their group,
    groups die and empty groups get filled (mixture.hpp:84-89,108-119)."""
    n, k = 96, 48
    orc, gpu = both(config, n, k, 20.0, 0.5, empty=empty, mode=mode)
    seed = 99
    st = ol.oracle().orc_rng_seed(seed)
    for sweep in range(4):
        for b in range(0, n, 16):
            orc.gibbs_batch(b, b + 16, st, sweep * n)
        gpu.sweep(0, n, 16, seed, draw_base=sweep * n)
        assert_same_state(orc, gpu, "dynamic sweep %d" % sweep)
    st2 = orc.gibbs_sequential(0, n, st)
    assert gpu.sweep_sequential(0, n, st) == st2
    assert_same_state(orc, gpu, "dynamic sequential")


@pytest.mark.parametrize("config,dim", [("dd", 16), ("dd_skew", 24), ("bb", None),
                                        ("dpd", 40), ("bnb", None)])
@pytest.mark.parametrize("stream", [0, 2])
@pytest.mark.parametrize("empty", [1, 4])
def test_device_side_normalisation_under_group_churn(config, dim, stream,
                                                     empty):
    """Sweeps that stay on the value-sorted path normalise the group set on
    the device and are queued without a host round trip: in the one launch
    that also builds the next batch's tables (k_vs_tables, the default) or by
    k_normalise + k_batch_finish (fused_tables = 0).  Many small groups and a
    large alpha make groups die and empty groups fill in nearly every batch,
    several per batch: both device-normalised engines, the host-normalised
    engine and the oracle agree bit for bit, ids included."""
    from distributions_amd import engine
    n, k = 6000, 900
    osh, gsh, vals, assign = workloads.make(config, n, k, dim=dim)
    orc = ol.OracleMixture(30.0, 0.6, osh)
    orc.init_from_assignments(vals, assign, k, empty)
    engines = []
    for normalise, fused in ((1, 1), (0, 0), (1, 0)):
        gpu = engine.Gibbs(30.0, 0.6, gsh)
        gpu.set_option("value_sorted", 2)
        gpu.set_option("value_stream", stream)
        gpu.set_option("device_normalise", normalise)
        gpu.set_option("fused_tables", fused)
        # (tables: one engine through k_vs_narrow, the others through the
        # 128-row tiles)
        gpu.set_option("narrow_tiles", 2 * fused)
        gpu.set_option("debug.narrow_read_ahead", 4)
        gpu.load_rows(vals, assign, k, empty)
        engines.append(gpu)
    seed = 4242
    st = ol.oracle().orc_rng_seed(seed)
    sizes = []
    for sweep, batch in enumerate([1500, 1000, 6000, 700]):
        for b in range(0, n, batch):
            orc.gibbs_batch(b, min(n, b + batch), st, sweep * n)
        sizes.append(len(orc))
        for gpu in engines:
            gpu.sweep(0, n, batch, seed, draw_base=sweep * n)
            assert_same_state(orc, gpu, "%s sweep %d" % (config, sweep))
            for slot in range(0, len(orc), 37):
                assert gpu.core.packed_to_global(slot) == \
                    orc.packed_to_global(slot)
            assert gpu.core.global_size() == orc.global_size()
    assert len(set(sizes)) > 1                    # the group set did change
    assert engines[0].core.debug_counts()["device_normalised"] > 0
    assert engines[1].core.debug_counts()["device_normalised"] == 0
    assert engines[2].core.debug_counts()["device_normalised"] > 0
    # (the table-free kernel keeps the separate launches)
    fused = engines[0].core.debug_counts()["fused_batches"]
    assert (fused > 0) == (stream == 0)
    assert engines[2].core.debug_counts()["fused_batches"] == 0
    # and the host-driven paths pick the state up where the device left it
    st2 = orc.gibbs_sequential(0, 300, st)
    assert engines[0].sweep_sequential(0, 300, st) == st2
    assert_same_state(orc, engines[0], "sequential after device sweeps")


def test_randomised_configurations():
    """random feature lists, group counts, hyper-parameters, batch sizes and
    seeds: every sweep bit-exact against the oracle"""
    from distributions_amd import engine
    rng = np.random.default_rng(31337)
    L = ol.oracle()
    for trial in range(24):
        n = int(rng.integers(200, 9000))
        k = int(rng.integers(1, 40))
        empty = int(rng.integers(1, 4))
        alpha = float(rng.choice([0.1, 1.0, 7.5]))
        d = float(rng.choice([0.0, 0.2, 0.7]))
        feats_o, feats_g, vals = [], [], []
        for _ in range(int(rng.integers(1, 4))):
            kind = rng.choice(["dd", "bb", "gp", "nich"])
            if kind == "dd":
                dim = int(rng.integers(2, 40))
                alphas = [float(a) for a in rng.uniform(0.2, 3.0, dim)]
                feats_o.append(ol.make_shared(ol.DD, alphas=alphas))
                feats_g.append(engine.dd_shared(alphas))
                vals.append(rng.integers(0, dim, n).astype(np.uint32))
            elif kind == "bb":
                a, b = float(rng.uniform(0.3, 4)), float(rng.uniform(0.3, 4))
                feats_o.append(ol.make_shared(ol.BB, alpha=a, beta=b))
                feats_g.append(engine.bb_shared(a, b))
                vals.append((rng.random(n) < 0.4).astype(np.uint32))
            elif kind == "gp":
                a = float(rng.choice([0.3, 0.5, 1.0, 1.7, 2.4, 4.0]))
                ib = float(rng.uniform(0.3, 3))
                feats_o.append(ol.make_shared(ol.GP, alpha=a, inv_beta=ib))
                feats_g.append(engine.gp_shared(a, ib))
                vals.append(rng.poisson(float(rng.uniform(0.5, 30)), n).astype(
                    np.uint32))
            else:
                p = [float(rng.normal()), float(rng.uniform(0.3, 3)),
                     float(rng.uniform(0.3, 3)), float(rng.uniform(0.5, 5))]
                feats_o.append(ol.make_shared(ol.NICH, mu=p[0], kappa=p[1],
                                              sigmasq=p[2], nu=p[3]))
                feats_g.append(engine.nich_shared(*p))
                vals.append((rng.normal(size=n) * 3).astype(np.float32))
        assign = rng.integers(0, k, n).astype(np.uint32)
        # every initial group must be non-empty
        assign[:k] = np.arange(k)
        orc = ol.OracleMixture(alpha, d, feats_o)
        orc.init_from_assignments(vals, assign, k, empty)
        gpu = engine.Gibbs(alpha, d, feats_g)
        gpu.set_option("value_sorted", int(rng.choice([0, 1, 2])))
        gpu.set_option("value_stream", int(rng.choice([0, 1, 2])))
        gpu.set_option("narrow_tiles", int(rng.choice([0, 1, 2])))
        gpu.set_option("debug.narrow_read_ahead", int(rng.choice([0, 4, 8])))
        gpu.set_option("debug.stream_scratch", int(rng.choice([0, 1])))
        gpu.set_option("device_normalise", int(rng.choice([0, 1, 2])))
        gpu.load_rows(vals, assign, k, empty)
        seed = int(rng.integers(1, 2 ** 31))
        st = L.orc_rng_seed(seed)
        batch = int(rng.choice([64, 777, 4096, n]))
        for sweep in range(2):
            for b in range(0, n, batch):
                orc.gibbs_batch(b, min(n, b + batch), st, sweep * n)
            gpu.sweep(0, n, batch, seed, draw_base=sweep * n)
            assert_same_state(orc, gpu, "trial %d sweep %d" % (trial, sweep))


@pytest.mark.parametrize("config", ["dd", "bb", "gp", "dd_bb_gp", "nich",
                                    "gp_nich"])
def test_delta_all_reduce_path_single_rank_nccl(config):
    """the multi-GPU code path (integer statistic deltas, RCCL all-reduce,
    apply, row exchange + ordered replay of the NICH / GP log_prod statistics,
    lock-step normalisation) driven with ONE rank on the real backend: must
    equal the direct path bit for bit"""
    import os
    import torch
    import torch.distributed as dist
    from distributions_amd import engine, _core
    n, k = 20000, 20
    osh, gsh, vals, assign = workloads.make(config, n, k)
    orc = ol.OracleMixture(1.0, 0.2, osh)
    orc.init_from_assignments(vals, assign, k, 2)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        cols = [torch.from_numpy(ol.value_words(s.kind, v).view(np.int32)).to(dev)
                for s, v in zip(osh, vals)]
        a = torch.from_numpy(assign.view(np.int32)).to(dev)
        gpu = engine.Gibbs(1.0, 0.2, gsh)
        gpu.load_rows_torch(cols, a, k, 2)
        sharded = engine.ShardedGibbs(gpu.core, n, 0, device=dev,
                                      force_collective=True, columns=cols,
                                      assign_packed=a)
        sharded.sync_initial_stats()
        assert_same_state(orc, gpu, "%s after the initial replay" % config)
        st = ol.oracle().orc_rng_seed(5)
        for sweep in range(2):
            for b in range(0, n, 6000):
                orc.gibbs_batch(b, min(n, b + 6000), st, sweep * n)
            sharded.sweep(6000, _core.rng_seed(5), draw_base=sweep * n)
            torch.cuda.synchronize()
            assert_same_state(orc, gpu, "%s delta sweep %d" % (config, sweep))
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("alpha", [0.3, 0.5, 1.25, 2.2])
def test_gamma_poisson_small_arguments_use_libm_values(alpha):
    """fast_lgamma(y < 2.5) is glibc's lgammaf in the reference
    (special.hpp:121-123); with a non-integer alpha and near-empty groups the
    device must reproduce those values (gp_lgamma's lookup), not an
    approximation: scores and assignments stay bit-exact."""
    from distributions_amd import engine
    rng = np.random.default_rng(3)
    n, k = 600, 150
    vals = [rng.integers(0, 3, n).astype(np.uint32)]
    assign = (np.arange(n) % k).astype(np.uint32)
    osh = [ol.make_shared(ol.GP, alpha=alpha, inv_beta=0.7)]
    orc = ol.OracleMixture(2.0, 0.3, osh)
    orc.init_from_assignments(vals, assign, k, 2)
    for mode in (0, 2):
        gpu = engine.Gibbs(2.0, 0.3, [engine.gp_shared(alpha, 0.7)])
        gpu.set_option("value_sorted", mode)
        gpu.load_rows(vals, assign, k, 2)
        for row in [0, 7, 599]:
            g = int(ol.oracle().orc_mix_global_to_packed(orc.h,
                                                        int(orc.assign[row])))
            want = orc.row_scores(row, g)
            got = gpu.row_scores(row)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    st = ol.oracle().orc_rng_seed(11)
    for sweep in range(3):
        for b in range(0, n, 200):
            orc.gibbs_batch(b, b + 200, st, sweep * n)
        gpu.sweep(0, n, 200, 11, draw_base=sweep * n)
        assert_same_state(orc, gpu, "gp alpha=%g sweep %d" % (alpha, sweep))


@pytest.mark.parametrize("alpha,beta,r", [(0.3, 0.45, 1), (1.25, 0.2, 1),
                                          (0.6, 1.1, 2), (1.0, 1.0, 1)])
def test_beta_negative_binomial_small_arguments_use_libm_values(alpha, beta,
                                                               r):
    """BetaNegativeBinomial hands fast_lgamma arguments below 2.5 whenever a
    group is empty or nearly so (bnb.hpp:200-223); the device must return
    glibc's lgammaf there (the registered table of special.h), so scores and
    assignments stay bit-exact for non-integer hyper-parameters."""
    from distributions_amd import engine
    rng = np.random.default_rng(5)
    n, k = 600, 150
    vals = [rng.integers(0, 3, n).astype(np.uint32)]
    assign = (np.arange(n) % k).astype(np.uint32)
    osh = [ol.make_shared(ol.BNB, alpha=alpha, beta=beta, r=r)]
    orc = ol.OracleMixture(2.0, 0.3, osh)
    orc.init_from_assignments(vals, assign, k, 2)
    for mode in (0, 2):
        gpu = engine.Gibbs(2.0, 0.3, [engine.bnb_shared(alpha, beta, r)])
        gpu.set_option("value_sorted", mode)
        gpu.load_rows(vals, assign, k, 2)
        for row in [0, 7, 599]:
            g = int(ol.oracle().orc_mix_global_to_packed(orc.h,
                                                        int(orc.assign[row])))
            want = orc.row_scores(row, g)
            got = gpu.row_scores(row)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    st = ol.oracle().orc_rng_seed(11)
    for sweep in range(3):
        for b in range(0, n, 200):
            orc.gibbs_batch(b, b + 200, st, sweep * n)
        gpu.sweep(0, n, 200, 11, draw_base=sweep * n)
        assert_same_state(orc, gpu, "bnb %g %g %d sweep %d" % (
            alpha, beta, r, sweep))


@pytest.mark.parametrize("config,mode", [("dd", 0), ("dd", 2), ("bb", 2),
                                         ("gp_nich", 0), ("bnb", 2)])
@pytest.mark.parametrize("slack", [0, 1000])
def test_low_entropy_clustering_sweeps_bit_exact(config, mode, slack):
    """the row update under Clustering::LowEntropy (clustering.hpp:245-331)
    scored through the generic MixtureDriver (mixture.hpp:124-141) instead of
    PitmanYor's cached driver: batches, sequential chain, groups appearing
    and vanishing; dataset_size == N (no size correction) and > N"""
    from distributions_amd import engine
    L = ol.oracle()
    L.orc_mix_set_low_entropy.restype = None
    L.orc_mix_set_low_entropy.argtypes = [ctypes.c_void_p, ctypes.c_int]
    n, k = 3000, 40
    osh, gsh, vals, assign = workloads.make(config, n, k)
    orc = ol.OracleMixture(1.0, 0.0, osh)
    L.orc_mix_set_low_entropy(orc.h, n + slack)
    orc.init_from_assignments(vals, assign, k, 2)
    gpu = engine.Gibbs(0.0, 0.0, gsh, dataset_size=n + slack)
    gpu.set_option("value_sorted", mode)
    gpu.load_rows(vals, assign, k, 2)
    for row in [0, 17, n - 1]:
        g = int(L.orc_mix_global_to_packed(orc.h, int(orc.assign[row])))
        want = orc.row_scores(row, g)
        got = gpu.row_scores(row)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    seed = 31
    st = L.orc_rng_seed(seed)
    for sweep in range(2):
        for b in range(0, n, 700):
            orc.gibbs_batch(b, min(n, b + 700), st, sweep * n)
        gpu.sweep(0, n, 700, seed, draw_base=sweep * n)
        assert_same_state(orc, gpu, "low entropy %s sweep %d" % (config, sweep))
    # and the sequential chain (batches of one)
    state = orc.gibbs_sequential(0, 300, st)
    assert gpu.sweep_sequential(0, 300, st) == state
    assert_same_state(orc, gpu, "low entropy %s sequential" % config)


def test_changing_the_batch_tiling_between_sweeps():
    """value-sorted batches cache their range's assignments by position; a
    different tiling of the same rows (other batch size, the sequential
    chain) must not leave a stale cache behind"""
    from distributions_amd import engine
    n, k = 6000, 24
    osh, gsh, vals, assign = workloads.make("dd", n, k)
    orc = ol.OracleMixture(1.0, 0.2, osh)
    orc.init_from_assignments(vals, assign, k, 1)
    gpu = engine.Gibbs(1.0, 0.2, gsh)
    gpu.set_option("value_sorted", 2)
    gpu.load_rows(vals, assign, k, 1)
    L = ol.oracle()
    seed = 9
    st = L.orc_rng_seed(seed)
    draw = 0
    for step, batch in enumerate([1500, 1000, None, 1500, 6000, 1500]):
        if batch is None:
            state = orc.gibbs_sequential(100, 400, st)
            assert gpu.sweep_sequential(100, 400, st) == state
        else:
            for b in range(0, n, batch):
                orc.gibbs_batch(b, min(n, b + batch), st, draw)
            gpu.sweep(0, n, batch, seed, draw_base=draw)
            draw += n
        assert_same_state(orc, gpu, "tiling step %d" % step)


def test_nich_with_tiny_nu_uses_libm_values_for_empty_groups():
    """fast_lgamma_nu(nu < 1/16) is two glibc lgammaf calls in the reference
    (special.hpp:226-229); a NormalInverseChiSq prior with nu = 0.01 reaches
    it for every group without members"""
    from distributions_amd import engine
    rng = np.random.default_rng(8)
    n, k = 500, 30
    vals = [rng.normal(0, 2, n).astype(np.float32)]
    assign = (np.arange(n) % k).astype(np.uint32)
    osh = [ol.make_shared(ol.NICH, mu=0.1, kappa=0.5, sigmasq=1.5, nu=0.01)]
    orc = ol.OracleMixture(1.0, 0.1, osh)
    orc.init_from_assignments(vals, assign, k, 3)
    gpu = engine.Gibbs(1.0, 0.1, [engine.nich_shared(0.1, 0.5, 1.5, 0.01)])
    gpu.load_rows(vals, assign, k, 3)
    L = ol.oracle()
    for row in [0, 250, 499]:
        g = int(L.orc_mix_global_to_packed(orc.h, int(orc.assign[row])))
        want = orc.row_scores(row, g)
        got = gpu.row_scores(row)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    st = L.orc_rng_seed(3)
    for sweep in range(2):
        for b in range(0, n, 125):
            orc.gibbs_batch(b, b + 125, st, sweep * n)
        gpu.sweep(0, n, 125, 3, draw_base=sweep * n)
        assert_same_state(orc, gpu, "tiny nu sweep %d" % sweep)


def test_independent_engines_on_their_own_streams():
    """dist_set_stream: two host threads, each with its own HIP stream and
    engine, run sequential chains concurrently; both reproduce the oracle"""
    import threading
    import torch
    from distributions_amd import _core, engine
    n, k = 1500, 20
    results = {}

    def worker(tag, config, seed):
        stream = torch.cuda.Stream()
        _core.set_stream(stream.cuda_stream)
        try:
            osh, gsh, vals, assign = workloads.make(config, n, k)
            gpu = engine.Gibbs(1.0, 0.2, gsh)
            gpu.load_rows(vals, assign, k, 1)
            st = gpu.sweep_sequential(0, n, _core.rng_seed(seed))
            gpu.sweep(0, n, 500, seed, draw_base=0)
            results[tag] = (st, gpu.assignments().copy(), gpu.counts().copy())
        except Exception as e:   # noqa: BLE001
            results[tag] = e
        finally:
            _core.set_stream(0)

    jobs = [("a", "dd", 3), ("b", "gp_nich", 4)]
    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    L = ol.oracle()
    for tag, config, seed in jobs:
        assert not isinstance(results[tag], Exception), results[tag]
        osh, gsh, vals, assign = workloads.make(config, n, k)
        orc = ol.OracleMixture(1.0, 0.2, osh)
        orc.init_from_assignments(vals, assign, k, 1)
        st = L.orc_rng_seed(seed)
        state = orc.gibbs_sequential(0, n, st)
        for b in range(0, n, 500):
            orc.gibbs_batch(b, b + 500, st, 0)
        got_state, got_assign, got_counts = results[tag]
        assert got_state == state
        np.testing.assert_array_equal(got_assign, orc.assign)
        np.testing.assert_array_equal(got_counts, orc.counts())


@pytest.mark.parametrize("config", ["dd", "nich", "gp_nich", "dd_bb_gp",
                                    "bnb"])
@pytest.mark.parametrize("prior_only", [False, True])
@pytest.mark.parametrize("empty", [1, 3])
def test_sequential_init_bit_exact(config, prior_only, empty):
    """a chain started from an empty mixture -- rows added
    one at a time, score -> sample -> add (examples/mixture/main.py:227-232,
    265-270) -- then a sequential and a batched sweep from that state."""
    from distributions_amd import engine
    n = 4096
    osh, gsh, vals, _ = workloads.make(config, n, 8)
    orc = ol.OracleMixture(2.0, 0.3, osh)
    orc.init_empty(vals, empty)
    gpu = engine.Gibbs(2.0, 0.3, gsh)
    gpu.load_rows_unassigned(vals, empty)
    with pytest.raises(RuntimeError, match="init_sequential first"):
        gpu.sweep(0, n, 1024, 1)
    st = ol.oracle().orc_rng_seed(31337)
    # in two stretches: the second resumes where the first stopped
    a = orc.init_sequential(0, 1000, st, prior_only)
    b = gpu.init_sequential(0, 1000, st, prior_only)
    assert a == b
    with pytest.raises(RuntimeError, match="in order"):
        gpu.init_sequential(5, 10, st)
    a = orc.init_sequential(1000, n, a, prior_only)
    b = gpu.init_sequential(1000, n, b, prior_only)
    assert a == b
    assert_same_state(orc, gpu, "%s after the init" % config)
    assert len(orc) > empty          # groups were created
    assert orc.gibbs_sequential(0, 500, a) == gpu.sweep_sequential(0, 500, b)
    seed = 99
    s2 = ol.oracle().orc_rng_seed(seed)
    for bb in range(0, n, 2048):
        orc.gibbs_batch(bb, bb + 2048, s2, 0)
    gpu.sweep(0, n, 2048, seed)
    assert_same_state(orc, gpu, "%s after a sweep" % config)


@pytest.mark.parametrize("config,dim,big_values", [
    ("dd", 2, False), ("dd", 16, False), ("dd_skew", 16, False),
    ("dpd", 3, False), ("bb", None, False), ("bnb", None, False),
    ("bnb", None, True), ("gp", None, True)])
def test_fused_batches_with_values_of_several_apply_chunks(config, dim,
                                                           big_values):
    """A fused batch's chunks sample the rows they were handed inside
    k_vs_apply, in the launch in which sibling chunks add up their moves.  A
    value with more than 5120 rows has several chunks: their count cells are
    left to k_vs_reduce (a handed-over row must see the cell as the batch
    found it), and batches with rows beyond the value table of a count model
    (BetaNegativeBinomial / GammaPoisson values > 255: sums changed in place)
    keep the separate launches.  Many small groups and a large alpha: rows
    alone in their group -- the handed-over kind -- in every batch.  Bit for
    bit against the oracle, and validate() after every sweep."""
    from distributions_amd import engine
    n, k = 30000, 700
    osh, gsh, vals, assign = workloads.make(config, n, k, dim=dim, seed=11)
    if big_values:      # rows the 256-entry tables do not cover, > one chunk
        rs = np.random.default_rng(2)
        where = rs.choice(n, 5000, replace=False)
        vals[0][where] = rs.integers(256, 4000, where.size).astype(np.uint32)
    orc = ol.OracleMixture(25.0, 0.5, osh)
    orc.init_from_assignments(vals, assign, k, 2)
    gpu = engine.Gibbs(25.0, 0.5, gsh)
    gpu.set_option("value_sorted", 2)
    gpu.set_option("device_normalise", 1)
    gpu.set_option("fused_tables", 1)
    gpu.load_rows(vals, assign, k, 2)
    seed = 777
    st = ol.oracle().orc_rng_seed(seed)
    for sweep, batch in enumerate([n, 15000, n]):
        for b in range(0, n, batch):
            orc.gibbs_batch(b, min(n, b + batch), st, sweep * n)
        gpu.sweep(0, n, batch, seed, draw_base=sweep * n)
        assert gpu.validate()["code"] == 0
        assert_same_state(orc, gpu, "%s sweep %d" % (config, sweep))
    counts = gpu.core.debug_counts()
    # (GammaPoisson's log-product is a float statistic: replayed in row order
    # between batches, its runs are not device-normalised at all)
    assert (counts["device_normalised"] > 0) == (config != "gp")
    assert (counts["fused_batches"] > 0) == (not big_values)


@pytest.mark.parametrize("config,dim", [("dd", 64), ("dpd", 40), ("bb", None),
                                        ("bnb", None)])
@pytest.mark.parametrize("overlap", [1, 0])
def test_fused_chunks_with_one_or_two_handed_over_rows(config, dim, overlap):
    """Where a chunk was handed only a few rows (the usual case at full size:
    a handful of rows alone in their group per value) its last waves sample
    them, one each, WHILE the others add up the moves of the rest
    (vs_deferred_row, k_vs_apply; debug.apply_overlap 0: before, as chunks
    with many do).  Moderate alpha: every batch has such rows, few per chunk.
    Bit for bit against the oracle either way, validate() after each sweep."""
    from distributions_amd import engine
    n, k = 60000, 300
    osh, gsh, vals, assign = workloads.make(config, n, k, dim=dim, seed=5)
    orc = ol.OracleMixture(3.0, 0.3, osh)
    orc.init_from_assignments(vals, assign, k, 2)
    gpu = engine.Gibbs(3.0, 0.3, gsh)
    gpu.set_option("value_sorted", 2)
    gpu.set_option("device_normalise", 1)
    gpu.set_option("fused_tables", 1)
    gpu.set_option("debug.apply_overlap", overlap)
    gpu.load_rows(vals, assign, k, 2)
    seed = 4242
    st = ol.oracle().orc_rng_seed(seed)
    for sweep, batch in enumerate([n, 20000, n, 30000]):
        for b in range(0, n, batch):
            orc.gibbs_batch(b, min(n, b + batch), st, sweep * n)
        gpu.sweep(0, n, batch, seed, draw_base=sweep * n)
        assert gpu.validate()["code"] == 0
        assert_same_state(orc, gpu, "%s sweep %d" % (config, sweep))
    counts = gpu.core.debug_counts()
    assert counts["fused_batches"] > 0
