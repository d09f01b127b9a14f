"""DirichletDiscrete -- mirror of distributions/lp/models/dd.pyx (+ _dd.pyx)."""
import numpy as np

from ... import _core
from ._base import (SharedBase, GroupBase, MixtureBase, SamplerBase,
                    sample_group_with)

NAME = 'DirichletDiscrete'
EXAMPLES = [
    {
        'shared': {'alphas': [0.5, 0.5, 0.5, 0.5]},
        'values': [0, 1, 0, 2, 0, 1, 0],
    },
    {
        'shared': {'alphas': [1.0, 4.0]},
        'values': [0, 1, 1, 1, 1, 0, 1],
    },
    {
        'shared': {'alphas': [2.0 / n for n in range(1, 21)]},
        'values': list(range(20)),
    },
]
Value = int


class Shared(SharedBase):
    FIELDS = ('alphas',)

    def load(self, raw):                       # dd.pyx:53-59
        alphas = [float(a) for a in raw['alphas']]
        self._params = _core.SharedParams.make(_core.KIND_DD, alphas=alphas)

    def dump(self):                            # dd.pyx:61-66
        return {'alphas': [float(a) for a in self.params.alphas]}


class Group(GroupBase):
    def __init__(self):
        GroupBase.__init__(self)
        self.dim = 0                           # dd.pyx:86-89

    def init(self, shared):
        self.dim = shared.params.dim
        GroupBase.init(self, shared)

    def _after_load(self):
        self.dim = len(self.words) - 1

    def load(self, raw):                       # dd.pyx:92-99
        counts = [int(c) for c in raw['counts']]
        self.dim = len(counts)
        self.words = np.array([sum(counts)] + counts).astype(np.uint32)

    def dump(self):                            # dd.pyx:101-106
        return {'counts': [int(c) for c in
                           self.words[1:1 + self.dim].astype(np.int32)]}

    def merge(self, shared, source):           # dd.hpp:151-158 (counts only)
        self.words[1:] += source.words[1:]

    def protobuf_load(self, message):
        self.load({'counts': list(message.counts)})

    def protobuf_dump(self, message):
        message.Clear()
        message.counts.extend(self.dump()['counts'])


class Mixture(MixtureBase):
    GROUP = Group


class Sampler(SamplerBase):                    # lp/models/_dd.pyx:71-81
    pass


def sample_group(shared, size):                # lp/models/_dd.pyx:141-151
    return sample_group_with(Group, shared, size)
