"""GammaPoisson -- mirror of distributions/lp/models/gp.pyx (+ _gp.pyx)."""
import numpy as np

from ... import _core
from ._base import (SharedBase, GroupBase, MixtureBase, SamplerBase,
                    sample_group_with)

NAME = 'GammaPoisson'
EXAMPLES = [
    {
        'shared': {'alpha': 1., 'inv_beta': 1.},
        'values': [0, 1, 2, 3, 4, 5, 6, 7, 8, 9],
    },
]
Value = int


class Shared(SharedBase):
    FIELDS = ('alpha', 'inv_beta')

    def load(self, raw):
        self._params = _core.SharedParams.make(
            _core.KIND_GP, p=(float(raw['alpha']), float(raw['inv_beta'])))

    def dump(self):
        p = self.params.p
        return {'alpha': p[0], 'inv_beta': p[1]}


class Group(GroupBase):
    def _after_load(self):
        pass

    def load(self, raw):
        w = np.zeros(3, np.uint32)
        w[0], w[1] = int(raw['count']), int(raw['sum'])
        w[2:3] = np.array([raw['log_prod']], np.float32).view(np.uint32)
        self.words = w

    def dump(self):
        return {'count': int(self.words[0]), 'sum': int(self.words[1]),
                'log_prod': float(self.words[2:3].view(np.float32)[0])}

    def merge(self, shared, source):           # gp.hpp:137-144
        a, b = self.dump(), source.dump()
        self.load({'count': a['count'] + b['count'], 'sum': a['sum'] + b['sum'],
                   'log_prod': np.float32(a['log_prod'])
                   + np.float32(b['log_prod'])})

    def protobuf_load(self, message):
        self.load({'count': message.count, 'sum': message.sum,
                   'log_prod': message.log_prod})

    def protobuf_dump(self, message):
        d = self.dump()
        message.count, message.sum = d['count'], d['sum']
        message.log_prod = d['log_prod']


class Mixture(MixtureBase):
    GROUP = Group


class Sampler(SamplerBase):                    # lp/models/_dd.pyx:71-81
    pass


def sample_group(shared, size):                # lp/models/_dd.pyx:141-151
    return sample_group_with(Group, shared, size)
