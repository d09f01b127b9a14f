"""The lp samplers -- Group.sample_value, Sampler(init, eval), sample_group --
restating distributions/tests/test_models.py:374-421 (goodness of fit of the
samples to exp(score_value) / exp(score_data)).  Host-side code (numpy
variates seeded from the global engine): runs without a GPU."""
import math

import numpy as np
import pytest
from scipy import stats

from distributions_amd.lp import random as lp_random
from distributions_amd.lp.models import bb, bnb, dd, gp, nich

SAMPLE_COUNT = 20000
MIN_GOODNESS_OF_FIT = 1e-3


def examples(mod):
    return [(mod, i) for i in range(len(mod.EXAMPLES))]


def chi2(samples, probs_dict):
    keys = sorted(probs_dict, key=repr)
    counts = np.array([samples.count(k) for k in keys], float)
    probs = np.array([probs_dict[k] for k in keys], float)
    n = len(samples)
    keep = probs * n >= 5
    obs = np.append(counts[keep], n - counts[keep].sum())
    exp = np.append(probs[keep] * n, n - (probs[keep] * n).sum())
    if exp[-1] < 1e-9:
        obs, exp = obs[:-1], exp[:-1]
    return stats.chisquare(obs, exp * obs.sum() / exp.sum()).pvalue


@pytest.mark.parametrize("mod,i", examples(dd) + examples(bb) + examples(gp)
                         + examples(bnb))
def test_discrete_sample_value_matches_score_value(mod, i):
    lp_random.seed(0)
    ex = mod.EXAMPLES[i]
    shared = mod.Shared.from_dict(ex['shared'])
    for values in ([], ex['values']):
        group = mod.Group.from_values(shared, values)
        samples = [group.sample_value(shared) for _ in range(SAMPLE_COUNT)]
        probs = {v: math.exp(group.score_value(shared, v))
                 for v in set(samples)}
        gof = chi2(samples, probs)
        print(mod.NAME, len(values), "gof", gof)
        assert gof > MIN_GOODNESS_OF_FIT


@pytest.mark.parametrize("i", range(len(nich.EXAMPLES)))
def test_nich_sample_value_follows_the_posterior_predictive(i):
    """the predictive is Student-t (nich.hpp:239-260 is its log density)"""
    lp_random.seed(0)
    ex = nich.EXAMPLES[i]
    shared = nich.Shared.from_dict(ex['shared'])
    group = nich.Group.from_values(shared, ex['values'])
    samples = np.array([group.sample_value(shared) for _ in range(5000)])
    p = shared.dump()
    x = np.array(ex['values'], float)
    n = len(x)
    kn = p['kappa'] + n
    mun = (p['kappa'] * p['mu'] + x.sum()) / kn
    nun = p['nu'] + n
    ss = ((x - x.mean()) ** 2).sum() if n else 0.0
    sign = (p['nu'] * p['sigmasq'] + ss
            + n * p['kappa'] * (p['mu'] - (x.mean() if n else 0)) ** 2 / kn
            ) / nun
    scale = math.sqrt(sign * (kn + 1) / kn)
    assert stats.kstest(samples, 't', args=(nun, mun, scale)).pvalue > 1e-3
    # ... and exp(score_value) is that density (to the fast functions' error)
    v = float(samples[0])
    assert abs(math.exp(group.score_value(shared, v))
               - stats.t.pdf(v, nun, mun, scale)) < 2e-3


@pytest.mark.parametrize("mod", [dd, bb, gp])
def test_sample_group_matches_score_data(mod):
    lp_random.seed(0)
    shared = mod.Shared.from_dict(mod.EXAMPLES[0]['shared'])
    samples, probs = [], {}
    for _ in range(6000):
        values = tuple(mod.sample_group(shared, 2))
        samples.append(values)
        if values not in probs:
            group = mod.Group.from_values(shared, list(values))
            probs[values] = math.exp(group.score_data(shared))
    assert chi2(samples, probs) > MIN_GOODNESS_OF_FIT


def test_sampler_object_and_seeding():
    shared = gp.Shared.from_dict(gp.EXAMPLES[0]['shared'])
    group = gp.Group.from_values(shared, gp.EXAMPLES[0]['values'])
    draws = []
    for _ in range(2):
        lp_random.seed(7)
        sampler = gp.Sampler()
        sampler.init(shared, group)
        draws.append([sampler.eval(shared) for _ in range(10)])
    assert draws[0] == draws[1]
