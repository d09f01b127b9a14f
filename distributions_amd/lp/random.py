"""Mirror of distributions/lp/random.pyx for the row-update path: the global
engine, sample_unif01 and discrete sampling from scores.

The reference keeps ONE process-global std::default_random_engine
(distributions/rng.py:37-47, global_rng.pyx:32-33) seeded through
distributions.hp.random.seed (hp/random.pyx:52-53); `seed()` here plays that
role.  The engine state is one 32-bit word (minstd_rand0).
"""
import numpy as np

from .. import _core


class RNG(object):                              # lp/random.pyx:66-80
    def __init__(self, seed=1):
        self.state = _core.rng_seed(seed)

    def seed(self, seed):
        self.state = _core.rng_seed(seed)

    def __call__(self):
        self.state = _core.rng_next(self.state)
        return self.state

    def copy(self):
        other = RNG()
        other.state = self.state
        return other


_global_rng = RNG()


def get_rng():
    return _global_rng


def seed(s):
    _global_rng.seed(s)


def random():
    """sample_unif01 (random.hpp:47-50)"""
    u, _global_rng.state = _core.rng_unif01(_global_rng.state)
    return u


def sample_discrete(probs):
    """sample_discrete (random.hpp:300-313) on normalised probabilities"""
    probs = np.ascontiguousarray(probs, np.float32)
    t = np.float32(random())
    for i in range(len(probs) - 1):
        t = np.float32(t - probs[i])
        if t < 0:
            return i
    return len(probs) - 1


def sample_from_scores(scores):
    """sample_from_scores (random.hpp:387-392): scores are not modified"""
    s = np.array(scores, np.float32)
    sample, _global_rng.state = _core.sample_from_scores_overwrite(
        _global_rng.state, s)
    return sample


def sample_prob_from_scores(scores):
    """sample_prob_from_scores_overwrite (random.hpp:369-376)"""
    s = np.array(scores, np.float32)
    total = np.float32(_core.scores_to_likelihoods(s))
    sample, _global_rng.state = _core.sample_from_likelihoods(
        _global_rng.state, s, total)
    return sample, float(s[sample] / total)


def prob_from_scores(sample, scores):
    """exp(score_from_scores_overwrite) (random.cc:108-128); consumes one
    engine step like the reference (SYNCHRONIZE_ENTROPY_FOR_UNIT_TESTING)"""
    s = np.array(scores, np.float32)
    lse = _core.log_sum_exp(s)
    random()
    return float(np.exp(np.float64(s[sample]) - lse))


def log_sum_exp(scores):
    """log_sum_exp (random.cc:77-92)"""
    return _core.log_sum_exp(np.asarray(scores, np.float32))
