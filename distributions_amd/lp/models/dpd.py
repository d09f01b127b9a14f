"""DirichletProcessDiscrete -- mirror of distributions/lp/models/dpd.pyx.

Values are kept in a dense remap: the Shared's `betas` dict {value: beta}
fixes an order value -> 0..V-1; OTHER is 0xFFFFFFFF (dpd.hpp:56).  The
stick-breaking side of Shared (add_value creating new values, realize;
dpd.hpp:66-101) is outside the row-update path.
"""
import numpy as np

from ... import _core
from ._base import SharedBase, GroupBase, MixtureBase

NAME = 'DirichletProcessDiscrete'
EXAMPLES = [
    {
        'shared': {
            'gamma': 0.5, 'alpha': 0.5,
            'betas': {0: 0.25, 1: 0.25, 2: 0.25, 3: 0.25},
            'counts': {0: 1, 1: 2, 2: 4, 3: 1},
        },
        'values': [0, 1, 0, 2, 0, 1, 0, 3],
    },
]
Value = int
OTHER = 0xFFFFFFFF


class Shared(SharedBase):
    def load(self, raw):
        self.gamma = float(raw.get('gamma', 1.0))
        self.alpha = float(raw['alpha'])
        self.values = sorted(int(v) for v in raw['betas'])
        self.index = {v: i for i, v in enumerate(self.values)}
        betas = [float(raw['betas'][v] if v in raw['betas']
                       else raw['betas'][str(v)]) for v in self.values]
        self.counts = dict(raw.get('counts', {}))
        self.beta0 = max(0.0, 1.0 - float(np.sum(betas, dtype=np.float64)))
        self._params = _core.SharedParams.make(
            _core.KIND_DPD, p=(self.alpha, self.beta0), betas=betas)

    def dump(self):
        betas = self.params.betas
        return {'gamma': self.gamma, 'alpha': self.alpha,
                'betas': {v: float(betas[i]) for v, i in self.index.items()},
                'counts': dict(self.counts)}

    def remap(self, value):
        return OTHER if value == OTHER else self.index[int(value)]


class Group(GroupBase):
    def _after_load(self):
        pass

    @staticmethod
    def _word(shared, value):
        return shared.remap(value)

    def add_value(self, shared, value):
        shared.params.group_add_value(self.words, shared.remap(value))

    def remove_value(self, shared, value):
        shared.params.group_remove_value(self.words, shared.remap(value))

    def score_value(self, shared, value):
        return shared.params.group_score_value(self.words,
                                               shared.remap(value))

    def merge(self, shared, source):           # sparse.hpp:163-168 (as built)
        self.words += source.words


class Mixture(MixtureBase):
    GROUP = Group

    def add_value(self, shared, groupid, value):
        self._handle(shared).add_value(groupid, shared.remap(value))

    def remove_value(self, shared, groupid, value):
        self._handle(shared).remove_value(groupid, shared.remap(value))

    def score_value_group(self, shared, groupid, value):
        return self._handle(shared).score_value_group(groupid,
                                                      shared.remap(value))

    def score_value(self, shared, value, scores_accum):
        assert len(scores_accum) == len(self), "scores_accum != len(mixture)"
        self._handle(shared).score_value(shared.remap(value), scores_accum)
