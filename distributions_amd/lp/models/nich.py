"""NormalInverseChiSq -- mirror of distributions/lp/models/nich.pyx (+ _nich.pyx)."""
import numpy as np

from ... import _core
from ._base import (SharedBase, GroupBase, MixtureBase, SamplerBase,
                    sample_group_with)

NAME = 'NormalInverseChiSq'
EXAMPLES = [
    {
        'shared': {'mu': 0., 'kappa': 1., 'sigmasq': 1., 'nu': 1.},
        'values': [-4.0, -2.0, 0.0, 1.0, 2.0, 3.0, 4.0, 5.0],
    },
]
Value = float


class Shared(SharedBase):
    FIELDS = ('mu', 'kappa', 'sigmasq', 'nu')

    def load(self, raw):
        self._params = _core.SharedParams.make(
            _core.KIND_NICH, p=tuple(float(raw[k]) for k in self.FIELDS))

    def dump(self):
        return dict(zip(self.FIELDS, self.params.p))


class Group(GroupBase):
    def _after_load(self):
        pass

    def load(self, raw):
        w = np.zeros(3, np.uint32)
        w[0] = int(raw['count'])
        w[1:3] = np.array([raw['mean'], raw['count_times_variance']],
                          np.float32).view(np.uint32)
        self.words = w

    def dump(self):
        f = self.words[1:3].view(np.float32)
        return {'count': int(self.words[0]), 'mean': float(f[0]),
                'count_times_variance': float(f[1])}

    def merge(self, shared, source):           # nich.hpp:167-179
        a, b = self.dump(), source.dump()
        f32 = np.float32
        total = a['count'] + b['count']
        delta = f32(b['mean']) - f32(a['mean'])
        source_part = f32(b['count']) / f32(total)
        cross_part = f32(a['count']) * source_part
        self.load({
            'count': total,
            'mean': f32(a['mean']) + source_part * delta,
            'count_times_variance': f32(a['count_times_variance'])
            + (f32(b['count_times_variance']) + cross_part * (delta * delta)),
        })

    def protobuf_load(self, message):
        self.load({'count': message.count, 'mean': message.mean,
                   'count_times_variance': message.count_times_variance})

    def protobuf_dump(self, message):
        d = self.dump()
        message.count, message.mean = d['count'], d['mean']
        message.count_times_variance = d['count_times_variance']


class Mixture(MixtureBase):
    GROUP = Group


class Sampler(SamplerBase):                    # lp/models/_dd.pyx:71-81
    pass


def sample_group(shared, size):                # lp/models/_dd.pyx:141-151
    return sample_group_with(Group, shared, size)
