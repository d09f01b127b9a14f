"""Edge cases of the row update on the GPU against the oracle: empty and
ragged inputs, degenerate group sets, extreme values (the reference's own
tests cover the analogous corners of its API: empty mixtures, single groups,
out-of-range group ids)."""
import numpy as np
import pytest

import oracle_lib as ol
from test_gpu_sweep import assert_same_state

pytestmark = pytest.mark.gpu


def run(osh, gsh, vals, assign, k, empty, batches, alpha=1.0, d=0.2, mode=None,
        sweeps=2, seed=17):
    from distributions_amd import engine
    n = len(assign)
    orc = ol.OracleMixture(alpha, d, osh)
    orc.init_from_assignments(vals, assign, k, empty)
    gpu = engine.Gibbs(alpha, d, gsh)
    if mode is not None:
        gpu.set_option("value_sorted", mode)
    gpu.load_rows(vals, assign, k, empty)
    assert_same_state(orc, gpu, "after load")
    st = ol.oracle().orc_rng_seed(seed)
    for sweep in range(sweeps):
        batch = batches[sweep % len(batches)]
        for b in range(0, n, batch):
            orc.gibbs_batch(b, min(n, b + batch), st, sweep * n)
        gpu.sweep(0, n, batch, seed, draw_base=sweep * n)
        assert_same_state(orc, gpu, "sweep %d batch %d" % (sweep, batch))
    return orc, gpu


def dd(dim, alpha=0.5):
    from distributions_amd import engine
    return ([ol.make_shared(ol.DD, alphas=[alpha] * dim)],
            [engine.dd_shared([alpha] * dim)])


@pytest.mark.parametrize("mode", [0, 2])
def test_no_rows(mode):
    """an engine without rows: every entry point is a no-op"""
    from distributions_amd import engine
    osh, gsh = dd(4)
    gpu = engine.Gibbs(1.0, 0.2, gsh)
    gpu.set_option("value_sorted", mode)
    empty = np.zeros(0, np.uint32)
    gpu.load_rows([empty], empty, 3, 1)
    gpu.sweep(0, 0, 10, 1, draw_base=0)
    assert gpu.sweep_sequential(0, 0, 123) == 123
    assert list(gpu.counts()) == [0, 0, 0, 0]
    assert gpu.assignments().size == 0


@pytest.mark.parametrize("mode", [0, 2])
def test_one_row(mode):
    osh, gsh = dd(3)
    run(osh, gsh, [np.array([2], np.uint32)], np.array([0], np.uint32), 1, 1,
        [1], mode=mode, sweeps=3)


@pytest.mark.parametrize("mode", [0, 2])
def test_one_group_and_ragged_batches(mode):
    """every row starts in the same group; batch sizes that do not divide n"""
    rng = np.random.default_rng(2)
    n = 1000
    osh, gsh = dd(5)
    vals = [rng.integers(0, 5, n).astype(np.uint32)]
    run(osh, gsh, vals, np.zeros(n, np.uint32), 1, 1, [333, 7, 1000], mode=mode,
        sweeps=3)


@pytest.mark.parametrize("mode", [0, 2])
def test_every_row_alone(mode):
    """n groups of one row: every removal empties a group (mixture.hpp:108-119)
    and the group set collapses and regrows"""
    rng = np.random.default_rng(3)
    n = 300
    osh, gsh = dd(4)
    vals = [rng.integers(0, 4, n).astype(np.uint32)]
    run(osh, gsh, vals, np.arange(n, dtype=np.uint32), n, 2, [64, 300],
        mode=mode, alpha=5.0, d=0.5, sweeps=3)


@pytest.mark.parametrize("mode", [0, 2])
def test_widest_dirichlet_discrete_with_unused_values(mode):
    """dim = 256 (the model's maximum, dd.hpp:50), most values never seen"""
    rng = np.random.default_rng(4)
    n, k = 3000, 12
    osh, gsh = dd(256, alpha=0.1)
    vals = [rng.choice([0, 1, 128, 254, 255], n).astype(np.uint32)]
    run(osh, gsh, vals, (np.arange(n) % k).astype(np.uint32), k, 1, [1000],
        mode=mode)


@pytest.mark.parametrize("mode", [0, 2])
def test_gamma_poisson_counts_beyond_the_value_table(mode):
    """counts far above the per-value table of the value-sorted kernel (and
    above fast_log_factorial's 64-entry table, special.hpp:208-214) take the
    handed-over path"""
    from distributions_amd import engine
    rng = np.random.default_rng(5)
    n, k = 6000, 10
    v = rng.poisson(4.0, n).astype(np.uint32)
    v[rng.integers(0, n, 40)] = rng.integers(300, 100000, 40)
    osh = [ol.make_shared(ol.GP, alpha=2.0, inv_beta=0.5)]
    gsh = [engine.gp_shared(2.0, 0.5)]
    run(osh, gsh, [v], (np.arange(n) % k).astype(np.uint32), k, 1, [2000],
        mode=mode)


@pytest.mark.parametrize("mode", [0, 2])
def test_beta_negative_binomial_counts_beyond_the_value_table(mode):
    from distributions_amd import engine
    rng = np.random.default_rng(7)
    n, k = 6000, 10
    v = rng.negative_binomial(3, 0.4, n).astype(np.uint32)
    v[rng.integers(0, n, 40)] = rng.integers(300, 50000, 40)
    osh = [ol.make_shared(ol.BNB, alpha=1.5, beta=0.75, r=3)]
    gsh = [engine.bnb_shared(1.5, 0.75, 3)]
    run(osh, gsh, [v], (np.arange(n) % k).astype(np.uint32), k, 1, [2000],
        mode=mode)


def test_nich_wide_dynamic_range():
    """values spanning twelve orders of magnitude: the Welford statistics and
    the Student-t scores stay finite and bit-identical"""
    from distributions_amd import engine
    rng = np.random.default_rng(6)
    n, k = 2000, 8
    x = (rng.normal(size=n) * 10.0 ** rng.integers(-6, 6, n)).astype(np.float32)
    osh = [ol.make_shared(ol.NICH, mu=0.0, kappa=0.1, sigmasq=2.0, nu=1.5)]
    gsh = [engine.nich_shared(0.0, 0.1, 2.0, 1.5)]
    orc, gpu = run(osh, gsh, [x], (np.arange(n) % k).astype(np.uint32), k, 1,
                   [500])
    for g in range(len(gpu)):
        assert np.all(np.isfinite(gpu.get_group(0, g)[1:].view(np.float32)))


def test_bernoulli_constant_column():
    """BetaBernoulli with every value equal: one of the two value tables is
    never used"""
    from distributions_amd import engine
    n, k = 4000, 6
    osh = [ol.make_shared(ol.BB, alpha=0.3, beta=0.7)]
    gsh = [engine.bb_shared(0.3, 0.7)]
    for const in (0, 1):
        vals = [np.full(n, const, np.uint32)]
        for mode in (0, 2):
            run(osh, gsh, vals, (np.arange(n) % k).astype(np.uint32), k, 1,
                [1500], mode=mode)


def test_bad_arguments_are_reported():
    """errors surface as exceptions with the library's message, as the
    reference's DIST_ASSERTs do"""
    from distributions_amd import engine
    osh, gsh = dd(4)
    gpu = engine.Gibbs(1.0, 0.2, gsh)
    vals = [np.array([0, 1, 2, 3], np.uint32)]
    with pytest.raises(RuntimeError):
        gpu.load_rows([np.array([0, 1, 2, 9], np.uint32)],
                      np.zeros(4, np.uint32), 1, 1)          # value >= dim
    with pytest.raises(RuntimeError):
        gpu.load_rows(vals, np.array([0, 1, 2, 7], np.uint32), 3, 1)  # group id
    with pytest.raises(RuntimeError):
        gpu.load_rows(vals, np.zeros(4, np.uint32), 1, 0)     # no empty group
    gpu.load_rows(vals, np.zeros(4, np.uint32), 1, 1)
    with pytest.raises(RuntimeError):
        gpu.sweep(0, 5, 2, 1, draw_base=0)                    # beyond the rows
    with pytest.raises(RuntimeError):
        engine.Gibbs(-1.0, 0.2, gsh)                          # alpha <= 0
    with pytest.raises(RuntimeError):
        engine.Gibbs(1.0, 1.0, gsh)                           # d >= 1


def test_no_features_clustering_prior_only():
    """a mixture without component models: rows move under the clustering
    prior alone (the driver's scores, clustering.hpp:195-208)"""
    from distributions_amd import engine
    n, k = 800, 10
    assign = (np.arange(n) % k).astype(np.uint32)
    for batches in ([200], [1]):
        run([], [], [], assign, k, 1, batches, alpha=2.0, d=0.3, sweeps=2)


def test_the_widest_feature_list():
    """DIST_MAX_FEATURES = 8 features: eight categoricals fill the score
    program (two ops each); a ninth feature is refused"""
    from distributions_amd import engine
    rng = np.random.default_rng(11)
    n, k = 1500, 9
    osh = [ol.make_shared(ol.DD, alphas=[0.3, 0.6, 0.9]) for _ in range(8)]
    gsh = [engine.dd_shared([0.3, 0.6, 0.9]) for _ in range(8)]
    vals = [rng.integers(0, 3, n).astype(np.uint32) for _ in range(8)]
    run(osh, gsh, vals, (np.arange(n) % k).astype(np.uint32), k, 1, [400, 1])
    with pytest.raises(RuntimeError):
        engine.Gibbs(1.0, 0.2, gsh + [engine.bb_shared(1.0, 1.0)])
    # a list of mixed kinds at the limit
    osh = osh[:4] + [ol.make_shared(ol.BB, alpha=0.5, beta=0.5),
                     ol.make_shared(ol.GP, alpha=1.0, inv_beta=2.0),
                     ol.make_shared(ol.NICH, mu=0.0, kappa=1.0, sigmasq=1.0,
                                    nu=2.0),
                     ol.make_shared(ol.BNB, alpha=1.0, beta=2.0, r=2)]
    gsh = gsh[:4] + [engine.bb_shared(0.5, 0.5), engine.gp_shared(1.0, 2.0),
                     engine.nich_shared(0.0, 1.0, 1.0, 2.0),
                     engine.bnb_shared(1.0, 2.0, 2)]
    vals = vals[:4] + [(rng.random(n) < 0.5).astype(np.uint32),
                       rng.poisson(3.0, n).astype(np.uint32),
                       rng.normal(size=n).astype(np.float32),
                       rng.negative_binomial(2, 0.4, n).astype(np.uint32)]
    run(osh, gsh, vals, (np.arange(n) % k).astype(np.uint32), k, 1, [500])
