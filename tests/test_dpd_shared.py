"""DirichletProcessDiscrete::Shared's stick-breaking (dpd.hpp:59-124) through
the C ABI (dist_dpd_shared_*) and the lp mirror.  Host-side logic: runs
without a GPU.

Entropy is pinned to libstdc++ itself: tests/golden/variates_libstdcxx.json is
what std::gamma_distribution<double> over std::default_random_engine gave
(oracle/check_libstdcxx.cc variates; random.hpp:87-119), bit patterns and the
engine's next output after every draw.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
M = 2147483647


def golden():
    with open(os.path.join(HERE, "golden", "variates_libstdcxx.json")) as f:
        return json.load(f)


def bits(x):
    return int(np.float32(x).view(np.uint32))


def test_sample_gamma_and_beta_safe_are_libstdcxx_bit_for_bit():
    from distributions_amd import _core
    g = golden()
    for case in g["cases"]:
        state = _core.rng_seed(case["seed"])
        for want, nxt in zip(case["gamma_bits"], case["gamma_next"]):
            x, state = _core.sample_gamma(state, case["alpha"], case["beta"])
            assert bits(x) == want, case
            assert state * 16807 % M == nxt
        state = _core.rng_seed(case["seed"])
        for want, nxt in zip(case["beta_safe_bits"], case["beta_safe_next"]):
            x, state = _core.sample_beta_safe(state, case["alpha"],
                                              case["beta"], g["min_value"])
            assert bits(x) == want, case
            assert state * 16807 % M == nxt


@pytest.mark.parametrize("gamma", [0.5, 2.0, 5.0])
def test_add_value_breaks_the_stick_with_libstdcxx_draws(gamma):
    """dpd.hpp:66-74 with the draws libstdc++ gave: beta = beta0 *
    sample_beta_safe(rng, 1, gamma, MIN_BETA); beta0 = max(MIN_BETA, beta0 -
    beta), in binary32"""
    from distributions_amd import _core
    case = [c for c in golden()["cases"]
            if c["alpha"] == 1.0 and c["beta"] == gamma and c["seed"] == 7][0]
    shared = _core.DpdShared()
    shared.load(gamma, 0.5, [], [], [])
    state = _core.rng_seed(7)
    beta0 = np.float32(1.0)
    for i, b in enumerate(case["beta_safe_bits"][:20]):
        draw = np.array([b], np.uint32).view(np.float32)[0]
        beta = np.float32(beta0 * draw)
        beta0 = max(np.float32(1e-6), np.float32(beta0 - beta))
        state = shared.add_value(100 + i, state)
        state = shared.add_value(100 + i, state)      # a second row: no draw
        values, betas, counts = shared.dump()
        assert values[i] == 100 + i and counts[i] == 2
        assert bits(betas[i]) == bits(beta), i
        assert bits(shared.scalars()[2]) == bits(beta0), i
    assert state * 16807 % M == case["beta_safe_next"][19]


def test_remove_value_returns_the_beta_and_frees_the_slot():
    """dpd.hpp:76-83; the freed dense slot goes to the next new value"""
    from distributions_amd import _core
    shared = _core.DpdShared()
    shared.load(0.5, 0.5, [0, 7, 8], [0.25, 0.5, 0.125], [1, 2, 4])
    assert abs(shared.scalars()[2] - 0.125) < 1e-7
    assert [shared.slot(v) for v in (0, 7, 8)] == [0, 1, 2]
    assert shared.slot(0xFFFFFFFF) == 0xFFFFFFFF
    with pytest.raises(RuntimeError):
        shared.slot(9)
    with pytest.raises(RuntimeError):
        shared.remove_value(9)
    with pytest.raises(RuntimeError):
        shared.add_value(0xFFFFFFFF, 1)
    shared.remove_value(7)
    assert len(shared) == 3
    version = shared.version
    shared.remove_value(7)
    assert len(shared) == 2 and shared.version != version
    assert bits(shared.scalars()[2]) == bits(np.float32(0.125)
                                             + np.float32(0.5))
    values, betas, counts = shared.dump()
    assert values.tolist() == [0, 0xFFFFFFFF, 8] and betas[1] == 0
    state = shared.add_value(300, _core.rng_seed(3))
    assert shared.slot(300) == 1 and shared.slots == 3
    assert state != _core.rng_seed(3)
    view = shared.params()
    assert view.dim == 3 and view.p[0] == 0.5
    assert bits(view.p[1]) == bits(shared.scalars()[2])


def test_realize_spends_the_whole_stick():
    """dpd.hpp:85-101: new values from 1 + max(value) until beta0 <= 1e-4,
    the rest to one last value, beta0 = 0"""
    from distributions_amd import _core
    for gamma, loaded in ((0.5, [(0, 0.25), (7, 0.5)]), (2.0, []),
                          (30.0, [(4, 0.1)])):
        shared = _core.DpdShared()
        shared.load(gamma, 1.0, [v for v, _ in loaded], [b for _, b in loaded],
                    [1] * len(loaded))
        shared.realize(_core.rng_seed(11))
        values, betas, counts = shared.dump()
        assert shared.scalars()[2] == 0.0
        assert abs(float(betas.astype(np.float64).sum()) - 1.0) < 1e-5
        first_new = 1 + max([v for v, _ in loaded], default=-1)
        new = values[len(loaded):]
        assert new.tolist() == list(range(first_new, first_new + len(new)))
        assert (counts[len(loaded):] == 1).all() and (betas > 0).all()
        assert len(shared) <= 10000
    # 10 000 values at most (max_size): a stick that barely breaks
    shared = _core.DpdShared()
    shared.load(1e5, 1.0, [], [], [])
    shared.realize(_core.rng_seed(5))
    assert len(shared) == 10000 and shared.scalars()[2] == 0.0


def test_lp_shared_is_the_flavour_tests_shared():
    """what distributions/tests/test_model_flavors.py:61-75 does with a
    Shared: every value added, realize(), dump() -> from_dict"""
    from distributions_amd.lp import random as lprandom
    from distributions_amd.lp.models import dpd
    lprandom.seed(0)
    for EXAMPLE in dpd.EXAMPLES + [
            {'shared': {'gamma': 2.0, 'alpha': 2.0, 'betas': {},
                        'counts': {}},
             'values': [5, 4, 3, 2, 1, 0, 3, 2, 1]}]:
        shared = dpd.Shared.from_dict(EXAMPLE['shared'])
        before = dict(shared.dump()['counts'])
        for value in EXAMPLE['values']:
            shared.add_value(value)
        shared.realize()
        raw = shared.dump()
        assert shared.beta0 == 0.0
        assert abs(sum(raw['betas'].values()) - 1.0) < 1e-5
        for value in set(EXAMPLE['values']):
            assert raw['counts'][value] == before.get(value, 0) + \
                EXAMPLE['values'].count(value)
        again = dpd.Shared.from_dict(raw)
        assert again.dump()['betas'] == raw['betas']
        assert again.dump()['counts'] == raw['counts']
        for value in EXAMPLE['values']:
            shared.remove_value(value)
    assert lprandom.get_rng().state != 1    # the global engine moved
