"""The exact sequential chain on the device, structural steps included
(k_chains), and M of them in one launch (dist_gibbs_sweep_sequential_many:
BASELINE configs[3] read literally, "8 independent chains").  Every chain is
held to the oracle's restatement of the reference loop
(examples/mixture/main.py:236-244 over mixture.hpp:73-122, 361-398,
clustering.hpp:163-230): assignments, group order, sizes, every statistic,
the id maps and the entropy state, bit for bit."""
import numpy as np
import pytest

import oracle_lib as ol
import workloads

pytestmark = pytest.mark.gpu


def make_chain(config, n, k, alpha, d, empty, seed, dim=None, mode=2):
    from distributions_amd import engine
    osh, gsh, vals, assign = workloads.make(config, n, k, seed=seed, dim=dim)
    orc = ol.OracleMixture(alpha, d, osh)
    orc.init_from_assignments(vals, assign, k, empty)
    gpu = engine.Gibbs(alpha, d, gsh)
    gpu.set_option("debug.sequential_chain", mode)
    gpu.load_rows(vals, assign, k, empty)
    return orc, gpu


def assert_same(orc, gpu, what):
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
    for g in range(len(orc)):   # MixtureIdTracker (mixture.hpp:460-521)
        assert gpu.core.packed_to_global(g) == orc.packed_to_global(g), what
    assert gpu.core.global_size() == orc.global_size(), what


@pytest.mark.parametrize("config,dim", [
    ("dd", 16), ("dd_skew", 24), ("bb", None), ("gp", None), ("nich", None),
    ("bnb", None), ("dpd", 40), ("dpd_other", None), ("gp_nich", None),
    ("dd_bb_gp", None)])
@pytest.mark.parametrize("empty", [1, 3])
def test_one_chain_under_group_churn(config, dim, empty):
    """Few rows per group and a large alpha: rows alone in their group all
    the time (the group vanishes, the last one moves into its slot), empty
    groups filled (a fresh one is appended) -- every step of it inside the
    kernel, three sweeps without a host round trip in any."""
    n, k = 400, 150
    orc, gpu = make_chain(config, n, k, 20.0, 0.5, empty, 11, dim)
    st = ol.oracle().orc_rng_seed(5)
    for sweep in range(3):
        want = orc.gibbs_sequential(0, n, st)
        got = gpu.sweep_sequential(0, n, st)
        assert got == want
        st = want
        assert_same(orc, gpu, "%s sweep %d" % (config, sweep))
    # one launch per sweep: the chain never went back to the host
    assert gpu.core.chain_launches() == 3
    assert gpu.validate()["code"] == 0


def test_chain_agrees_with_the_older_paths():
    """the same chain through k_chains (2), round 3's kernel with the host at
    every structural step (1) and rows as batches of one (0)"""
    out = []
    for mode in (2, 1, 0):
        orc, gpu = make_chain("dd", 300, 40, 10.0, 0.3, 2, 3, 16, mode)
        st = gpu.sweep_sequential(0, 300, 777)
        out.append((st, gpu.assignments(), gpu.counts()))
    for other in out[1:]:
        assert other[0] == out[0][0]
        assert np.array_equal(other[1], out[0][1])
        assert np.array_equal(other[2], out[0][2])


def test_chain_that_runs_out_of_room_goes_on():
    """more groups founded in one call than a launch has room for (256): the
    kernel stops at a row boundary, the host reserves and relaunches"""
    n, k = 3000, 4
    orc, gpu = make_chain("dd", n, k, 2000.0, 0.9, 1, 17, 16)
    st = 4242
    want = orc.gibbs_sequential(0, n, st)
    assert gpu.sweep_sequential(0, n, st) == want
    assert len(orc) > k + 1 + 256
    assert gpu.core.chain_launches() > 1
    assert_same(orc, gpu, "out of room")


@pytest.mark.parametrize("m,n,k,dim", [(8, 3000, 1024, 256),
                                        (512, 2048, 1024, 256)])
def test_many_chains_in_one_launch(m, n, k, dim):
    """M chains of DirichletDiscrete(256) at K = 1024 (BASELINE configs[1]'s
    model, configs[3]'s "independent chains"): own rows (seed + i), own
    entropy, one launch; every one of them equals its oracle chain."""
    from distributions_amd import _core
    chains = [make_chain("dd", n, k, 1.0, 0.2, 1, 100 + i, dim)
              for i in range(m)]
    states = np.array([_core.rng_seed(9000 + i) for i in range(m)], np.uint32)
    got = _core.sweep_sequential_many([g.core for _, g in chains], 0, n,
                                      states)
    assert sum(g.core.chain_launches() for _, g in chains) <= 2
    # (the oracle: every chain at M = 8; a spread of them at M = 512, where
    # all of them would be 10^6 rows of CPU work)
    check = range(m) if m <= 8 else list(range(0, m, 37)) + [m - 1]
    for i in check:
        orc, gpu = chains[i]
        want = orc.gibbs_sequential(0, n, int(states[i]))
        assert int(got[i]) == want, i
        assert_same(orc, gpu, "chain %d" % i)
    # entropy states are distinct and every engine holds a valid state
    assert len(set(int(s) for s in got)) == m
    for _, gpu in chains[:16]:
        assert gpu.validate()["code"] == 0


def test_many_chains_of_mixed_rows():
    """BASELINE configs[2]'s feature list (GammaPoisson + NormalInverseChiSq:
    order-dependent float statistics) as 6 concurrent chains with churn"""
    from distributions_amd import _core
    m, n, k = 6, 500, 60
    chains = [make_chain("gp_nich", n, k, 8.0, 0.4, 2, 50 + i)
              for i in range(m)]
    states = np.array([_core.rng_seed(31 + i) for i in range(m)], np.uint32)
    for sweep in range(2):
        got = _core.sweep_sequential_many([g.core for _, g in chains], 0, n,
                                          states)
        for i, (orc, gpu) in enumerate(chains):
            assert int(got[i]) == orc.gibbs_sequential(0, n, int(states[i]))
            assert_same(orc, gpu, "chain %d sweep %d" % (i, sweep))
        states = got


@pytest.mark.parametrize("config", ["dd", "gp_nich"])
def test_chain_under_low_entropy_clustering_with_churn(config):
    """Clustering::LowEntropy through the generic MixtureDriver
    (clustering.hpp:245-331, mixture.hpp:124-141): the structural steps of the
    device chain under the other clustering model"""
    import ctypes
    from distributions_amd import engine
    L = ol.oracle()
    L.orc_mix_set_low_entropy.restype = None
    L.orc_mix_set_low_entropy.argtypes = [ctypes.c_void_p, ctypes.c_int]
    n, k = 400, 150
    osh, gsh, vals, assign = workloads.make(config, n, k, seed=23)
    orc = ol.OracleMixture(1.0, 0.0, osh)
    L.orc_mix_set_low_entropy(orc.h, n + 50)
    orc.init_from_assignments(vals, assign, k, 2)
    gpu = engine.Gibbs(0.0, 0.0, gsh, dataset_size=n + 50)
    gpu.load_rows(vals, assign, k, 2)
    st = 991
    for sweep in range(3):
        want = orc.gibbs_sequential(0, n, st)
        assert gpu.sweep_sequential(0, n, st) == want
        st = want
        assert_same(orc, gpu, "low entropy %s sweep %d" % (config, sweep))
    assert gpu.core.chain_launches() == 3
