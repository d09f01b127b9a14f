"""The HIP library against the REFERENCE'S OWN Python flavour: the lp mirror
takes the lp flavour's seat in distributions/tests/test_model_flavors.py:61-116
with the dbg flavour's answers read from tests/golden/dbg_models.json.gz
(made by tests/golden/make_dbg_goldens.py from the reference's
distributions/dbg/models/*.py, dbg/clustering.py).  Tolerance and comparison
are the reference's (tests/util.py:42,100-140); the allowances are the ones
tests/test_dbg_goldens.py documents for the oracle (the reference's own
lookup tables), nothing else.

  Group (host, one group)      every step: dump, score_value, score_data
  Mixture (HBM, the hot path)  K groups frozen at K points of the script, built
                               with Mixture.add_value / remove_value on the
                               device: score_value (accumulating, all groups
                               at once: mixture.hpp:416-425), score_value_group,
                               score_data
  Scorer                       dist_scorer_init / eval
  LowEntropy                   score_add_value / remove / counts / partition
"""
import numpy as np
import pytest

import dbg_fixtures as fx
from test_dbg_goldens import EPS_LOG, check_group_dump

pytestmark = pytest.mark.gpu


def module_of(name):
    import importlib
    return importlib.import_module("distributions_amd.lp.models." + name)


def lp_shared(name, scen):
    module = module_of(name)
    raw = scen["shared"]
    if name == "dpd":
        raw = dict(raw, betas={int(k): v for k, v in raw["betas"].items()},
                   counts={int(k): v for k, v in raw["counts"].items()})
    return module, module.Shared.from_dict(raw)


def allowance(name, scen, words, grid):
    from scipy.special import gammaln
    from distributions_amd import _core
    raw = scen["shared"]
    i32 = np.asarray(words).view(np.int32)
    if name == "nich":
        nu = float(np.float32(raw["nu"])) + float(i32[0])
        exact = gammaln(0.5 * nu + 0.5) - gammaln(0.5 * nu)
        err = abs(float(_core.vector_lgamma_nu(
            np.array([nu], np.float32))[0]) - exact)
        return np.full(len(grid), err + (0.5 * nu + 1.0) * EPS_LOG)
    if name == "gp":
        a = raw["alpha"] + float(i32[1])
        return np.array([(2.0 * a + float(x)) * EPS_LOG for x in grid])
    return np.zeros(len(grid))


def apply(obj, shared, step, *where):
    if step["op"] == "add":
        obj.add_value(shared, *where, step["value"])
    elif step["op"] == "remove":
        obj.remove_value(shared, *where, step["value"])


@pytest.mark.parametrize("name,index", fx.scenario_ids())
def test_group_follows_the_dbg_flavour(name, index):
    scen = fx.models()[name]["scenarios"][index]
    module, shared = lp_shared(name, scen)
    assert module.NAME == fx.models()[name]["NAME"]
    group = module.Group.from_values(shared)
    grid = scen["grid"]
    for t, step in enumerate(scen["steps"]):
        msg = "%s[%d] step %d" % (name, index, t)
        apply(group, shared, step)
        if "group" in step:
            check_group_dump(name, scen, np.asarray(group.words), step["group"],
                             msg)
        allow = allowance(name, scen, group.words, grid)
        got = [group.score_value(shared, v) for v in grid]
        fx.assert_close(got, step["score_value"], msg + " Group.score_value",
                        allow)
        fx.assert_close(group.score_data(shared), step["score_data"],
                        msg + " Group.score_data")


@pytest.mark.parametrize("name,index", fx.scenario_ids())
def test_mixture_on_the_device_follows_the_dbg_flavour(name, index):
    """what the row update reads: the device-resident value scorer"""
    scen = fx.models()[name]["scenarios"][index]
    module, shared = lp_shared(name, scen)
    steps = scen["steps"]
    grid = scen["grid"]
    K = min(12, len(steps))
    stops = sorted(set(np.linspace(0, len(steps) - 1, K).astype(int)))
    mixture = module.Mixture()
    for _ in stops:
        mixture.append(module.Group.from_values(shared))
    mixture.init(shared)
    for g, stop in enumerate(stops):
        for step in steps[1:stop + 1]:
            apply(mixture, shared, step, g)
    allow = np.stack([allowance(name, scen, mixture[g].words, grid)
                      for g in range(len(stops))])
    want = np.array([steps[s]["score_value"] for s in stops])
    rng = np.random.default_rng(index)
    for j, v in enumerate(grid):
        noise = rng.normal(size=len(stops)).astype(np.float32)
        acc = noise.copy()
        mixture.score_value(shared, v, acc)
        fx.assert_close(acc.astype(np.float64) - noise, want[:, j],
                        "%s[%d] Mixture.score_value(%r)" % (name, index, v),
                        allow[:, j] + 4e-6 * np.abs(noise))
        one = [mixture.score_value_group(shared, g, v)
               for g in range(len(stops))]
        fx.assert_close(one, want[:, j],
                        "%s[%d] Mixture.score_value_group" % (name, index),
                        allow[:, j])
    fx.assert_close(mixture.score_data(shared),
                    sum(steps[s]["score_data"] for s in stops),
                    "%s[%d] Mixture.score_data" % (name, index))
    # groups read back from HBM carry dbg's statistics
    for g, stop in enumerate(stops):
        dumps = [s for s in steps[:stop + 1] if "group" in s]
        if "group" in steps[stop]:
            check_group_dump(name, scen, np.asarray(mixture[g].words),
                             steps[stop]["group"], "group %d" % g)
        assert dumps


@pytest.mark.parametrize("name", ["dd", "bb", "gp", "nich", "bnb"])
def test_scorer_follows_the_dbg_flavour(name):
    """Model::Scorer (dd.hpp:224-245, gp.hpp:194-217, nich.hpp:232-259):
    init from a group, eval per value"""
    for index, scen in enumerate(fx.models()[name]["scenarios"]):
        module, shared = lp_shared(name, scen)
        group = module.Group.from_values(shared)
        grid = scen["grid"]
        for t, step in enumerate(scen["steps"]):
            apply(group, shared, step)
            if t % 7:
                continue
            state = shared.params.scorer_init(group.words)
            got = [shared.params.scorer_eval(
                state, module.Group._word(shared, v)) for v in grid]
            fx.assert_close(got, step["score_value"],
                            "%s[%d] step %d Scorer.eval" % (name, index, t),
                            allowance(name, scen, group.words, grid))


@pytest.mark.parametrize("case", range(6))
def test_low_entropy_follows_the_dbg_flavour(case):
    from distributions_amd.lp.clustering import LowEntropy
    c = fx.low_entropy()[case]
    model = LowEntropy(dataset_size=c["dataset_size"])
    for size, nonempty, sample, empties, want in c["score_add_value"]:
        fx.assert_close(model.score_add_value(size, nonempty, sample, empties),
                        want, "score_add_value(%d,%d,%d,%d)" % (
                            size, nonempty, sample, empties),
                        size * EPS_LOG if size <= 10000 else 0.0)
    for size, nonempty, sample, empties, want in c["score_remove_value"]:
        fx.assert_close(
            model.score_remove_value(size, nonempty, sample, empties), want,
            "score_remove_value(%d,%d)" % (size, sample),
            (size - 1) * EPS_LOG if size - 1 <= 10000 else 0.0)
    for counts, want in c["score_counts"]:
        fx.assert_close(model.score_counts(counts), want,
                        "score_counts(%r)" % (counts,))
    for n, want in c["log_partition_function"]:
        fx.assert_close(model.log_partition_function(n), want,
                        "log_partition_function(%d)" % n)
