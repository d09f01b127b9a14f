"""BetaNegativeBinomial -- mirror of distributions/lp/models/bnb.pyx
(+ _bnb.pyx) over include/distributions/models/bnb.hpp."""
import numpy as np

from ... import _core
from ._base import (SharedBase, GroupBase, MixtureBase, SamplerBase,
                    sample_group_with)

NAME = 'BetaNegativeBinomial'
EXAMPLES = [
    {
        'shared': {'alpha': 1., 'beta': 1., 'r': 1},
        'values': [0, 1, 2, 3, 4, 5, 6, 1, 2, 3, 4, 2, 3],
    },
]
Value = int


class Shared(SharedBase):
    FIELDS = ('alpha', 'beta', 'r')

    def load(self, raw):
        self._params = _core.SharedParams.make(
            _core.KIND_BNB,
            p=(float(raw['alpha']), float(raw['beta']), float(int(raw['r']))))

    def dump(self):
        p = self.params.p
        return {'alpha': p[0], 'beta': p[1], 'r': int(p[2])}


class Group(GroupBase):
    def _after_load(self):
        pass

    def load(self, raw):                       # bnb.pyx:71-73
        self.words = np.array([int(raw['count']), int(raw['sum'])], np.uint32)

    def dump(self):                            # bnb.pyx:75-79
        return {'count': int(self.words[0]), 'sum': int(self.words[1])}

    def merge(self, shared, source):           # bnb.hpp:133-139
        self.words = (self.words + source.words).astype(np.uint32)

    def protobuf_load(self, message):
        self.load({'count': message.count, 'sum': message.sum})

    def protobuf_dump(self, message):
        message.count, message.sum = int(self.words[0]), int(self.words[1])


class Mixture(MixtureBase):
    GROUP = Group


class Sampler(SamplerBase):                    # lp/models/_dd.pyx:71-81
    pass


def sample_group(shared, size):                # lp/models/_dd.pyx:141-151
    return sample_group_with(Group, shared, size)
