"""orc_mix_load_state: an oracle that ADOPTS a state -- group order, sizes,
statistics (float ones included), ids, assignments -- continues exactly like
the one that produced it.  (The full-size GPU tests rely on this to follow the
engine from a state only the engine has reached: sweep 2 at N = 10M.)"""
import numpy as np
import pytest

import oracle_lib as ol
import workloads


@pytest.mark.parametrize("config", ["dd", "gp_nich", "bb", "dpd", "bnb",
                                    "dd_bb_gp"])
def test_adopted_state_continues_identically(config):
    n, k = 6000, 40
    osh, _, vals, assign = workloads.make(config, n, k)
    a = ol.OracleMixture(1.0, 0.2, osh)
    a.init_from_assignments(vals, assign, k, 1)
    st = ol.oracle().orc_rng_seed(5)
    for b in range(0, n, 1500):          # groups die and are created here
        a.gibbs_batch(b, b + 1500, st, 0)
    assert len(a) != k + 1
    twin = ol.OracleMixture(1.0, 0.2, osh)
    twin.adopt(a, vals)
    assert len(twin) == len(a) and twin.global_size() == a.global_size()
    for sweep in (1, 2):
        for b in range(0, n, 1500):
            a.gibbs_batch(b, b + 1500, st, sweep * n)
            twin.gibbs_batch(b, b + 1500, st, sweep * n)
    np.testing.assert_array_equal(a.assign, twin.assign)
    np.testing.assert_array_equal(a.counts(), twin.counts())
    for f in range(a.F):
        for g in range(len(a)):
            np.testing.assert_array_equal(a.get_group(f, g),
                                          twin.get_group(f, g))
    assert a.gibbs_sequential(0, 500, st) == twin.gibbs_sequential(0, 500, st)
    np.testing.assert_array_equal(a.assign, twin.assign)
