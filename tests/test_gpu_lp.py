"""The reference's own tests for this path, restated against the lp mirror on
the GPU (distributions/tests/test_models.py:498-594,
distributions/tests/test_clustering.py:242-327,
distributions/tests/test_random.py:183-247), plus bit-exact checks of every
value against the oracle and the reference goldens."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
TOL = 1e-3   # distributions/tests/util.py:42
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def assert_close(a, b, tol=TOL, msg=""):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert np.all(np.abs(a - b) <= tol * (1.0 + np.abs(a) + np.abs(b))), (
        msg, a, b)


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def modules():
    from distributions_amd.lp.models import dd, bb, gp, nich, dpd, bnb
    return [dd, bb, gp, nich, dpd, bnb]


def examples():
    out = []
    for module in modules():
        for i, ex in enumerate(module.EXAMPLES):
            out.append(pytest.param(module, ex, id="%s-%d" % (module.NAME, i)))
    return out


def oracle_twin(module, shared):
    """the same Shared for the oracle"""
    p = shared.params
    kind = p.kind
    if kind == ol.DD:
        return ol.make_shared(ol.DD, alphas=p.alphas)
    if kind == ol.BB:
        return ol.make_shared(ol.BB, alpha=p.p[0], beta=p.p[1])
    if kind == ol.GP:
        return ol.make_shared(ol.GP, alpha=p.p[0], inv_beta=p.p[1])
    if kind == ol.NICH:
        return ol.make_shared(ol.NICH, mu=p.p[0], kappa=p.p[1],
                              sigmasq=p.p[2], nu=p.p[3])
    if kind == ol.BNB:
        return ol.make_shared(ol.BNB, alpha=p.p[0], beta=p.p[1], r=p.p[2])
    return ol.make_shared(ol.DPD, alpha=p.p[0], beta0=p.p[1], betas=p.betas)


@pytest.mark.parametrize("module,EXAMPLE", examples())
def test_mixture_runs(module, EXAMPLE):
    from distributions_amd.lp import random as lprandom
    lprandom.seed(0)
    shared = module.Shared.from_dict(EXAMPLE['shared'])
    values = EXAMPLE['values']
    mixture = module.Mixture()
    for value in values:
        shared.add_value(value)
        mixture.append(module.Group.from_values(shared, [value]))
    mixture.init(shared)
    groupids = []
    for value in values:
        scores = np.zeros(len(mixture), dtype=np.float32)
        mixture.score_value(shared, value, scores)
        groupid = lprandom.sample_from_scores(scores)
        mixture.add_value(shared, groupid, value)
        groupids.append(groupid)
    mixture.add_group(shared)
    assert len(mixture) == len(values) + 1
    for value, groupid in zip(values, groupids):
        mixture.remove_value(shared, groupid, value)
    mixture.remove_group(shared, 0)
    mixture.remove_group(shared, len(mixture) - 1)
    assert len(mixture) == len(values) - 1
    for value in values:
        scores = np.zeros(len(mixture), dtype=np.float32)
        mixture.score_value(shared, value, scores)
        groupid = lprandom.sample_from_scores(scores)
        mixture.add_value(shared, groupid, value)


@pytest.mark.parametrize("module,EXAMPLE", examples())
def test_mixture_score(module, EXAMPLE):
    """Mixture.score_value / score_value_group == per-group Group.score_value
    (TOL, like the reference) and == the oracle's MixtureSlave bit for bit."""
    L = ol.oracle()
    rng = np.random.default_rng(0)
    shared = module.Shared.from_dict(EXAMPLE['shared'])
    values = EXAMPLE['values']
    groups = [module.Group.from_values(shared, [value]) for value in values]
    mixture = module.Mixture()
    for group in groups:
        mixture.append(group)
    mixture.init(shared)

    osh = oracle_twin(module, shared)
    orc = ol.OracleMixture(1.0, 0.0, [osh])
    word = lambda v: module.Group._word(shared, v)   # noqa: E731
    for g, value in enumerate(values):
        L.orc_mix_slave_append_empty(orc.h, 0)
        L.orc_mix_slave_group_add_value(orc.h, 0, g, word(value))
    L.orc_mix_slave_init(orc.h, 0)

    def check_score_value(value):
        expected = [group.score_value(shared, value) for group in groups]
        actual = np.zeros(len(mixture), dtype=np.float32)
        noise = rng.normal(size=len(actual)).astype(np.float32)
        actual += noise
        want = noise.copy()
        mixture.score_value(shared, value, actual)
        L.orc_mix_slave_score_value(orc.h, 0, word(value), want)
        assert np.array_equal(bits(actual), bits(want)), "vs oracle"
        assert_close(actual - noise, expected, msg='score_value')
        another = [mixture.score_value_group(shared, i, value)
                   for i in range(len(groups))]
        want_g = [L.orc_mix_slave_score_value_group(orc.h, 0, i, word(value))
                  for i in range(len(groups))]
        assert np.array_equal(bits(another), bits(want_g)), "group vs oracle"
        assert_close(another, expected, msg='score_value_group')
        return actual - noise

    for value in values:
        check_score_value(value)
    groupids = []
    for value in values:
        scores = check_score_value(value)
        p = np.exp(scores - scores.max())
        groupid = int(rng.choice(len(p), p=p / p.sum()))
        groups[groupid].add_value(shared, value)
        mixture.add_value(shared, groupid, value)
        L.orc_mix_slave_add_value(orc.h, 0, groupid, word(value))
        groupids.append(groupid)
        np.testing.assert_array_equal(mixture[groupid].words,
                                      groups[groupid].words)
    for value, groupid in zip(values, groupids):
        groups[groupid].remove_value(shared, value)
        mixture.remove_value(shared, groupid, value)
        L.orc_mix_slave_remove_value(orc.h, 0, groupid, word(value))
        check_score_value(value)


def test_mixture_errors_are_runtime_errors():
    from distributions_amd.lp.models import dd
    shared = dd.Shared.from_dict(dd.EXAMPLES[0]['shared'])
    mixture = dd.Mixture()
    mixture.append(dd.Group.from_values(shared, [0]))
    mixture.init(shared)
    with pytest.raises(RuntimeError):
        mixture.add_value(shared, 5, 0)           # bad groupid
    with pytest.raises(RuntimeError):
        mixture.add_value(shared, 0, 4)           # value out of bounds
    with pytest.raises(AssertionError):
        mixture.score_value(shared, 0, np.zeros(3, np.float32))


@pytest.mark.parametrize("EXAMPLE", [
    {'alpha': 1., 'd': 0.}, {'alpha': 1., 'd': 0.1}, {'alpha': 1., 'd': 0.9},
    {'alpha': 10., 'd': 0.1}, {'alpha': 0.1, 'd': 0.1}])
def test_mixture_score_matches_score_add_value(EXAMPLE):
    from distributions_amd.lp.clustering import PitmanYor
    from distributions_amd.lp.mixture import MixtureIdTracker
    L = ol.oracle()
    rng = np.random.default_rng(1)
    model = PitmanYor()
    model.load(EXAMPLE)
    sample_count = 120
    nonempty_counts = [int(c) for c in rng.integers(1, 40, 9)]

    def check_counts(mixture, counts, empty_group_count):
        empty_groupids = frozenset(mixture.empty_groupids)
        assert len(empty_groupids) == empty_group_count
        for groupid in empty_groupids:
            assert counts[groupid] == 0
        np.testing.assert_array_equal(mixture.counts(), counts)

    def check_scores(mixture, counts, empty_group_count):
        sample_count = sum(counts)
        nonempty_group_count = len(counts) - empty_group_count
        expected = [model.score_add_value(group_size, nonempty_group_count,
                                          sample_count, empty_group_count)
                    for group_size in counts]
        want = [L.orc_py_score_add_value(model.alpha, model.d, c,
                                         nonempty_group_count, sample_count,
                                         empty_group_count) for c in counts]
        assert np.array_equal(bits(expected), bits(want))
        actual = rng.normal(size=len(counts)).astype(np.float32)
        mixture.score_value(model, actual)
        assert_close(actual, expected)
        return actual

    for empty_group_count in [1, 10]:
        counts = nonempty_counts + [0] * empty_group_count
        rng.shuffle(counts)
        counts = [int(c) for c in counts]
        mixture = PitmanYor.Mixture()
        id_tracker = MixtureIdTracker()
        mixture.init(model, counts)
        id_tracker.init(len(counts))
        orc = ol.OracleMixture(model.alpha, model.d, [])
        L.orc_mix_driver_init(orc.h, np.array(counts, np.int32), len(counts))

        def vs_oracle(scores):
            want = np.zeros(len(counts), np.float32)
            L.orc_mix_driver_score_value(orc.h, want)
            assert np.array_equal(bits(scores), bits(want))

        groupids = []
        for _ in range(sample_count):
            check_counts(mixture, counts, empty_group_count)
            scores = check_scores(mixture, counts, empty_group_count)
            vs_oracle(scores)
            p = np.exp(scores - scores.max())
            groupid = int(rng.choice(len(p), p=p / p.sum()))
            expected_group_added = (counts[groupid] == 0)
            counts[groupid] += 1
            assert mixture.add_value(model, groupid) == expected_group_added
            L.orc_mix_driver_add_value(orc.h, groupid)
            groupids.append(id_tracker.packed_to_global(groupid))
            if expected_group_added:
                id_tracker.add_group()
                counts.append(0)
        for global_groupid in groupids:
            groupid = id_tracker.global_to_packed(global_groupid)
            counts[groupid] -= 1
            expected_group_removed = (counts[groupid] == 0)
            assert (mixture.remove_value(model, groupid)
                    == expected_group_removed)
            L.orc_mix_driver_remove_value(orc.h, groupid)
            if expected_group_removed:
                id_tracker.remove_group(groupid)
                back = counts.pop()
                if groupid < len(counts):
                    counts[groupid] = back
            check_counts(mixture, counts, empty_group_count)
            vs_oracle(check_scores(mixture, counts, empty_group_count))


@pytest.mark.parametrize("name", ["fast_log", "fast_exp", "fast_lgamma",
                                  "fast_lgamma_nu", "fast_log_factorial"])
def test_special_functions_match_reference_goldens_on_gpu(name):
    """the device special functions against the REAL reference's outputs"""
    from distributions_amd.lp import special
    g = np.load(os.path.join(GOLD, "special_functions.npz"))
    x = g[name + "_in"]
    got = getattr(special, name)(x)
    bad = np.nonzero(bits(got) != g[name + "_out"])[0]
    assert bad.size == 0, (name, x[bad[:5]], got[bad[:5]])


def test_sampling_matches_oracle_and_reference_probe():
    from distributions_amd.lp import random as lprandom
    L = ol.oracle()
    lprandom.seed(1)
    got = [lprandom.sample_from_scores([-1, -2.5, .25, -.75, -3])
           for _ in range(8)]
    assert got == [0, 0, 2, 2, 2, 2, 0, 2]
    rng = np.random.default_rng(3)
    lprandom.seed(77)
    st = ctypes.c_uint32(L.orc_rng_seed(77))
    for size in [1, 2, 5, 64, 1024, 5000]:
        for _ in range(4):
            scores = (rng.normal(size=size) * 4).astype(np.float32)
            s2 = scores.copy()
            want = L.orc_sample_from_scores_overwrite(ctypes.byref(st), size,
                                                      s2)
            assert lprandom.sample_from_scores(scores) == want
            assert lprandom.get_rng().state == st.value
            assert np.float32(lprandom.log_sum_exp(scores)) == np.float32(
                L.orc_log_sum_exp(size, np.ascontiguousarray(scores)))
    # extreme spreads: flushed tails, ties
    for scores in [[0, -100, -88, -87.5, -50], [5, 5, 5, 5], [-1e30, 0.0]]:
        s = np.array(scores, np.float32)
        s2 = s.copy()
        want = L.orc_sample_from_scores_overwrite(ctypes.byref(st), s.size, s2)
        assert lprandom.sample_from_scores(s) == want


def test_prob_from_scores():
    from distributions_amd.lp import random as lprandom
    rng = np.random.default_rng(4)
    lprandom.seed(5)
    for size in range(1, 30):
        scores = rng.normal(size=size).tolist()
        sample, prob1 = lprandom.sample_prob_from_scores(scores)
        assert 0 <= sample < size
        prob2 = lprandom.prob_from_scores(sample, scores)
        assert_close(prob1, prob2)


@pytest.mark.parametrize("module,EXAMPLE", examples())
def test_mixture_score_data(module, EXAMPLE):
    """check_score_data of distributions/tests/test_models.py:559-562 on the
    GPU, plus the oracle: bit-exact (the device accumulates in the reference's
    float order; small lgamma arguments come from the registered glibc
    table), 1e-5 for DirichletProcessDiscrete"""
    L = ol.oracle()
    shared = module.Shared.from_dict(EXAMPLE['shared'])
    values = EXAMPLE['values']
    groups = [module.Group.from_values(shared, [value]) for value in values]
    mixture = module.Mixture()
    for group in groups:
        mixture.append(group)
    mixture.init(shared)
    osh = oracle_twin(module, shared)
    orc = ol.OracleMixture(1.0, 0.0, [osh])
    word = lambda v: module.Group._word(shared, v)   # noqa: E731
    for g, value in enumerate(values):
        L.orc_mix_slave_append_empty(orc.h, 0)
        L.orc_mix_slave_group_add_value(orc.h, 0, g, word(value))
    L.orc_mix_slave_init(orc.h, 0)

    def check():
        expected = sum(group.score_data(shared) for group in groups)
        actual = mixture.score_data(shared)
        assert_close(actual, expected, msg='score_data')
        want = L.orc_mix_slave_score_data(orc.h, 0)
        if module.NAME == 'DirichletProcessDiscrete':
            # the reference walks a hash map there: order-free sum, 1e-5
            assert abs(actual - want) <= 1e-5 * (1 + abs(want)), (actual, want)
        else:   # the reference's float accumulation order, bit for bit
            assert np.float32(actual) == np.float32(want), (actual, want)
        for g, group in enumerate(groups):
            wg = L.orc_group_score_data(ctypes.byref(osh), orc.get_group(0, g))
            assert np.float32(group.score_data(shared)) == np.float32(wg)

    check()
    for i, value in enumerate(values):
        g = (3 * i + 1) % len(groups)
        groups[g].add_value(shared, value)
        mixture.add_value(shared, g, value)
        L.orc_mix_slave_add_value(orc.h, 0, g, word(value))
        check()


def grid_candidates(module, raw):
    """a hyper-parameter grid around an EXAMPLE's shared, as dicts; neighbours
    differ in one or two entries (what DirichletDiscrete's incremental
    score_data_grid exploits, dd.hpp:268-281)"""
    import copy
    out = [copy.deepcopy(raw)]
    rng = np.random.default_rng(8)
    for step in range(7):
        d = copy.deepcopy(out[-1])
        if 'alphas' in d:
            for _ in range(1 + step % 2):
                i = int(rng.integers(len(d['alphas'])))
                d['alphas'][i] = float(d['alphas'][i]) * float(
                    rng.uniform(0.5, 2.0))
        else:
            keys = [k for k in sorted(d) if isinstance(d[k], float)
                    and k != 'beta0']
            k = keys[step % len(keys)]
            scale = float(rng.uniform(0.5, 2.0))
            d[k] = d[k] * scale if k != 'mu' else d[k] + scale
        out.append(d)
    return out


@pytest.mark.parametrize("module,EXAMPLE", examples())
def test_mixture_score_data_grid(module, EXAMPLE):
    """score_data_grid (mixture.hpp:433-438): every entry equals score_data
    under that candidate, and the oracle's restatement of the reference's
    loops (DirichletDiscrete's incremental one included) within 1e-5"""
    L = ol.oracle()
    L.orc_mix_slave_score_data_grid.restype = None
    L.orc_mix_slave_score_data_grid.argtypes = [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
        ctypes.c_void_p]
    shared = module.Shared.from_dict(EXAMPLE['shared'])
    values = EXAMPLE['values']
    word = lambda v: module.Group._word(shared, v)   # noqa: E731
    n_groups = 5
    mixture = module.Mixture()
    osh = oracle_twin(module, shared)
    orc = ol.OracleMixture(1.0, 0.0, [osh])
    for g in range(n_groups):
        mixture.append(module.Group.from_values(shared, []))
        L.orc_mix_slave_append_empty(orc.h, 0)
    mixture.init(shared)
    L.orc_mix_slave_init(orc.h, 0)
    for i, value in enumerate(values * 3):
        g = (5 * i + 2) % (n_groups - 1)        # the last group stays empty
        mixture.add_value(shared, g, value)
        L.orc_mix_slave_add_value(orc.h, 0, g, word(value))
    cands = [module.Shared.from_dict(d)
             for d in grid_candidates(module, shared.dump())]
    got = mixture.score_data_grid(cands)
    assert got.shape == (len(cands),) and got.dtype == np.float32
    twins = [oracle_twin(module, c) for c in cands]
    arr = (ol.Shared * len(twins))(*twins)
    want = np.zeros(len(twins), np.float32)
    L.orc_mix_slave_score_data_grid(orc.h, 0, ctypes.cast(arr, ctypes.c_void_p),
                                    len(twins), want.ctypes.data)
    for c, cand in enumerate(cands):
        single = module.Mixture()
        for g in range(n_groups):
            single.append(mixture[g])
        single.init(cand)
        assert abs(got[c] - single.score_data(cand)) <= 1e-6 * (
            1 + abs(got[c])), c
        if module.NAME == 'DirichletProcessDiscrete':
            assert abs(got[c] - want[c]) <= 1e-5 * (1 + abs(want[c])), (
                c, got[c], want[c])
        else:
            assert got[c] == want[c], (c, got[c], want[c])
    assert len(set(np.round(got, 3))) > 1      # the grid does vary the score


def test_py_score_counts_and_mixture_score_data():
    from distributions_amd.lp.clustering import PitmanYor
    L = ol.oracle()
    model = PitmanYor(alpha=1.0, d=0.2)
    got = model.score_counts([5, 3, 1, 0])
    assert abs(got - (-9.18923473)) < 5e-7
    rng = np.random.default_rng(12)
    for ex in PitmanYor.EXAMPLES:
        model = PitmanYor(**ex)
        counts = [int(c) for c in rng.integers(0, 50, 300)] + [0]
        want = L.orc_py_score_counts(model.alpha, model.d,
                                     np.array(counts, np.int32), len(counts))
        got = model.score_counts(counts)
        assert abs(got - want) <= 1e-6 * (1 + abs(want))
        mixture = PitmanYor.Mixture()
        mixture.init(model, counts)
        assert abs(mixture.score_data(model) - want) <= 1e-6 * (1 + abs(want))


def test_low_entropy_model_and_mixture():
    """LowEntropy through distributions_amd.lp: score_add_value and
    score_counts against the oracle (bit-exact / 1e-6), and the reference's
    test_mixture_score_matches_score_add_value (test_clustering.py:243-327)
    for LowEntropy.Mixture = MixtureDriver<LowEntropy>"""
    from distributions_amd.lp.clustering import LowEntropy
    from distributions_amd.lp.mixture import MixtureIdTracker
    from distributions_amd.lp import random as lprandom
    L = ol.oracle()
    i = ctypes.c_int
    L.orc_le_score_add_value.restype = ctypes.c_float
    L.orc_le_score_add_value.argtypes = [i, i, i, i, i]
    L.orc_le_score_counts.restype = ctypes.c_float
    L.orc_le_score_counts.argtypes = [i, ctypes.c_void_p, ctypes.c_size_t]
    rng = np.random.default_rng(21)
    lprandom.seed(3)
    for ex in LowEntropy.EXAMPLES + [{'dataset_size': 100000}]:
        model = LowEntropy(**ex)
        assert model.dump() == ex
        N = model.dataset_size
        for size, sample, empty in [(0, 0, 1), (0, N - 2, 3), (1, 1, 1),
                                    (3, N // 2, 2), (N - 1, N - 1, 1)]:
            if sample >= N or size > sample:
                continue
            want = L.orc_le_score_add_value(N, size, 1, sample, empty)
            got = model.score_add_value(size, 1, sample, empty)
            assert np.float32(got) == np.float32(want), (ex, size, sample)
            if size:
                assert model.score_remove_value(size + 1, 1, sample + 1,
                                                empty) == -got
        counts = [int(c) for c in rng.multinomial(
            min(N, 5000) // 2, np.ones(6) / 6)] + [0]
        c = np.ascontiguousarray(counts, np.int32)
        want = L.orc_le_score_counts(N, c.ctypes.data, c.size)
        got = model.score_counts(counts)
        assert abs(got - want) <= 1e-6 * (1 + abs(want)), (ex, got, want)

    # sample_assignments (clustering.cc:250-283): the engine's draws, bit for bit
    L.orc_le_sample_assignments.restype = None
    L.orc_le_sample_assignments.argtypes = [i, i, ctypes.c_void_p,
                                            ctypes.c_void_p]
    for dataset_size, size in [(5, 5), (100, 60), (1000, 400), (10 ** 6, 300)]:
        model = LowEntropy(dataset_size=dataset_size)
        lprandom.seed(77)
        got = model.sample_assignments(size)
        state = ctypes.c_uint32(L.orc_rng_seed(77))
        want = np.zeros(size, np.int32)
        L.orc_le_sample_assignments(dataset_size, size, ctypes.byref(state),
                                    want.ctypes.data)
        assert got == want.tolist()
        assert lprandom.get_rng().state == state.value

    model = LowEntropy(dataset_size=1000)
    nonempty_counts = [int(c) for c in rng.integers(1, 30, 7)]
    for empty_group_count in [1, 10]:
        counts = nonempty_counts + [0] * empty_group_count
        rng.shuffle(counts)
        counts = [int(c) for c in counts]
        mixture = LowEntropy.Mixture()
        id_tracker = MixtureIdTracker()
        mixture.init(model, counts)
        id_tracker.init(len(counts))

        def check(counts):
            empties = frozenset(mixture.empty_groupids)
            assert len(empties) == empty_group_count
            assert all(counts[g] == 0 for g in empties)
            expected = [model.score_add_value(
                size, len(counts) - empty_group_count, sum(counts),
                empty_group_count) for size in counts]
            actual = rng.normal(size=len(counts)).astype(np.float32)
            mixture.score_value(model, actual)
            assert np.array_equal(actual, np.float32(expected))
            assert abs(mixture.score_data(model) - model.score_counts(counts)
                       ) < 1e-4
            return actual

        groupids = []
        for _ in range(60):
            scores = check(counts)
            probs = np.exp(scores - scores.max())
            groupid = int(rng.choice(len(counts), p=probs / probs.sum()))
            added = mixture.add_value(model, groupid)
            assert added == (counts[groupid] == 0)
            counts[groupid] += 1
            groupids.append(id_tracker.packed_to_global(groupid))
            if added:
                id_tracker.add_group()
                counts.append(0)
        for global_id in groupids:
            groupid = id_tracker.global_to_packed(global_id)
            counts[groupid] -= 1
            removed = mixture.remove_value(model, groupid)
            assert removed == (counts[groupid] == 0)
            if removed:
                id_tracker.remove_group(groupid)
                back = counts.pop()
                if groupid < len(counts):
                    counts[groupid] = back
            check(counts)


def test_c1_benchmark_loop_dd16_k64():
    """benchmarks/mixture.cc:79-115 at BASELINE configs[0] (DirichletDiscrete
    dim = 16, K = 64), through distributions_amd.lp: groups filled with 4 * K
    values, then the timed loop's body -- remove the value from its group,
    score_value ACCUMULATING into a vector that is zeroed every 8 iterations
    (mixture.cc:104-113), add it back -- with the accumulated scores compared
    with the oracle's MixtureSlave bit for bit at every iteration."""
    from distributions_amd.lp.models import dd
    L = ol.oracle()
    K, dim, iters = 64, 16, 4096
    rng = np.random.default_rng(20240601)
    shared = dd.Shared.from_dict({"alphas": [0.5] * dim})
    values = rng.integers(0, dim, 4 * K).tolist()        # mixture.cc:93-100
    assignments = rng.integers(0, K, 4 * K).tolist()
    mixture = dd.Mixture()
    members = [[] for _ in range(K)]
    for v, g in zip(values, assignments):
        members[g].append(v)
    for g in range(K):
        mixture.append(dd.Group.from_values(shared, members[g]))
    mixture.init(shared)

    orc = ol.OracleMixture(1.0, 0.0, [ol.make_shared(ol.DD,
                                                     alphas=[0.5] * dim)])
    for g in range(K):
        L.orc_mix_slave_append_empty(orc.h, 0)
        for v in members[g]:
            L.orc_mix_slave_group_add_value(orc.h, 0, g, v)
    L.orc_mix_slave_init(orc.h, 0)

    scores = np.zeros(K, np.float32)
    want = np.zeros(K, np.float32)
    for i in range(iters):
        if i % 8 == 0:                                   # vector_zero
            scores[:] = 0
            want[:] = 0
        k = i % len(values)
        value, groupid = values[k], assignments[k]
        mixture.remove_value(shared, groupid, value)
        L.orc_mix_slave_remove_value(orc.h, 0, groupid, value)
        mixture.score_value(shared, value, scores)       # accumulates
        L.orc_mix_slave_score_value(orc.h, 0, value, want)
        mixture.add_value(shared, groupid, value)
        L.orc_mix_slave_add_value(orc.h, 0, groupid, value)
        assert np.array_equal(bits(scores), bits(want)), (i, scores, want)
    for g in range(K):
        assert mixture[g].dump() == dd.Group.from_values(
            shared, members[g]).dump()


def test_dpd_mixture_follows_a_shared_that_gains_and_loses_values():
    """the row loop of examples/mixture/main.py with a DirichletProcessDiscrete
    feature whose Shared starts EMPTY: shared.add_value(value) breaks the
    stick for every new value (dpd.hpp:66-74), remove_value gives it back
    (:76-83), and the device mixture is widened / rebuilt behind the lp
    surface.  After every phase: Mixture.score_value over every live value and
    OTHER == the oracle's MixtureSlave built from scratch on the Shared's
    current dense view with the same memberships, bit for bit."""
    from distributions_amd.lp import random as lprandom
    from distributions_amd.lp.models import dpd
    L = ol.oracle()
    lprandom.seed(11)
    rs = np.random.default_rng(5)
    shared = dpd.Shared.from_dict({'gamma': 3.0, 'alpha': 1.5, 'betas': {},
                                   'counts': {}})
    K = 5
    mixture = dpd.Mixture()
    members = [[] for _ in range(K)]

    def compare():
        p = shared.params
        osh = ol.make_shared(ol.DPD, alpha=p.p[0], beta0=p.p[1],
                             betas=p.betas)
        orc = ol.OracleMixture(1.0, 0.0, [osh])
        for g in range(K):
            L.orc_mix_slave_append_empty(orc.h, 0)
            for v in members[g]:
                L.orc_mix_slave_group_add_value(orc.h, 0, g, shared.remap(v))
        L.orc_mix_slave_init(orc.h, 0)
        live = [v for v in shared.values if v is not None]
        for v in live + [dpd.OTHER]:
            got = np.zeros(K, np.float32)
            want = np.zeros(K, np.float32)
            mixture.score_value(shared, v, got)
            L.orc_mix_slave_score_value(orc.h, 0, shared.remap(v), want)
            assert np.array_equal(bits(got), bits(want)), v
        for g in range(K):
            counts = mixture[g].dump()['counts']
            want = {}
            for v in members[g]:
                want[v] = want.get(v, 0) + 1
            assert counts == want

    first = [int(v) for v in rs.integers(0, 12, 40)]
    for g in range(K):
        shared.add_value(first[g])
        mixture.append(dpd.Group.from_values(shared, [first[g]]))
        members[g].append(first[g])
    mixture.init(shared)
    compare()
    for v in first[K:]:                       # new values appear mid-stream
        shared.add_value(v)
        scores = np.zeros(K, np.float32)
        mixture.score_value(shared, v, scores)
        g = lprandom.sample_from_scores(scores)
        mixture.add_value(shared, g, v)
        members[g].append(v)
    compare()
    rare = [v for v in set(first) if first.count(v) <= 3]
    for v in rare:                            # ... and vanish
        for g in range(K):
            while v in members[g]:
                mixture.remove_value(shared, g, v)
                members[g].remove(v)
                shared.remove_value(v)
    assert rare and all(v not in shared.dump()['betas'] for v in rare)
    assert shared.beta0 > 0
    compare()
    for v in (100, 101, 102, 100):            # freed slots are taken again
        shared.add_value(v)
        mixture.add_value(shared, 0, v)
        members[0].append(v)
    assert shared.params.dim <= 12
    compare()


@pytest.mark.parametrize("module,EXAMPLE", examples())
def test_mixture_score_values_is_score_value_in_one_launch(module, EXAMPLE):
    """dist_mixture_score_values (extension): a batch of values scored
    against all groups in one launch == one Mixture.score_value per value
    (mixture.hpp:416-425), bit for bit, accumulating like it"""
    rng = np.random.default_rng(3)
    shared = module.Shared.from_dict(EXAMPLE['shared'])
    values = EXAMPLE['values']
    mixture = module.Mixture()
    for value in values:
        shared.add_value(value)
    for value in values:
        mixture.append(module.Group.from_values(shared, [value]))
    mixture.init(shared)
    for g, value in enumerate(values):
        mixture.add_value(shared, (g + 2) % len(mixture), value)
    batch = [values[i] for i in rng.integers(0, len(values), 37)]
    noise = rng.normal(size=(len(batch), len(mixture))).astype(np.float32)
    got = noise.copy()
    mixture.score_values(shared, batch, got)
    for r, value in enumerate(batch):
        want = noise[r].copy()
        mixture.score_value(shared, value, want)
        assert np.array_equal(bits(got[r]), bits(want)), (r, value)
    mixture.validate(shared)
