"""BetaBernoulli -- mirror of distributions/lp/models/bb.pyx (+ _bb.pyx)."""
import numpy as np

from ... import _core
from ._base import (SharedBase, GroupBase, MixtureBase, SamplerBase,
                    sample_group_with)

NAME = 'BetaBernoulli'
EXAMPLES = [
    {
        'shared': {'alpha': 0.5, 'beta': 2.0},
        'values': [False, False, True, False, True, True, False, False],
    },
    {
        'shared': {'alpha': 10.5, 'beta': 0.5},
        'values': [False, False, False, False, False, False, False, True],
    },
]
Value = bool


class Shared(SharedBase):
    FIELDS = ('alpha', 'beta')

    def load(self, raw):
        self._params = _core.SharedParams.make(
            _core.KIND_BB, p=(float(raw['alpha']), float(raw['beta'])))

    def dump(self):
        p = self.params.p
        return {'alpha': p[0], 'beta': p[1]}


class Group(GroupBase):
    def _after_load(self):
        pass

    def load(self, raw):
        self.words = np.array([raw['heads'], raw['tails']]).astype(np.uint32)

    def dump(self):
        w = self.words.astype(np.int32)
        return {'heads': int(w[0]), 'tails': int(w[1])}

    def merge(self, shared, source):           # bb.hpp:124-130
        self.words += source.words

    def protobuf_load(self, message):
        self.load({'heads': message.heads, 'tails': message.tails})

    def protobuf_dump(self, message):
        d = self.dump()
        message.heads, message.tails = d['heads'], d['tails']


class Mixture(MixtureBase):
    GROUP = Group


class Sampler(SamplerBase):                    # lp/models/_dd.pyx:71-81
    pass


def sample_group(shared, size):                # lp/models/_dd.pyx:141-151
    return sample_group_with(Group, shared, size)
