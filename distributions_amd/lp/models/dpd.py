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

    def protobuf_load(self, message):          # dpd.pyx:105-120
        values = [int(v) for v in message.values]
        self.load({
            'gamma': message.gamma, 'alpha': message.alpha,
            'betas': dict(zip(values, (float(b) for b in message.betas))),
            'counts': dict(zip(values, (int(c) for c in message.counts))),
        })

    def protobuf_dump(self, message):          # dpd.pyx:122-134
        message.Clear()
        message.gamma = self.gamma
        message.alpha = self.alpha
        betas = self.params.betas
        for value in self.values:
            message.values.append(value)
            message.betas.append(float(betas[self.index[value]]))
            message.counts.append(int(self.counts.get(value, 0)))


class Group(GroupBase):
    """counts live densely in `words` once the group has met its Shared (the
    value -> index map is the Shared's); a group loaded from a dict or a
    message before that keeps the sparse {value: count} until then"""

    def __init__(self):
        GroupBase.__init__(self)
        self._values = None     # dense index -> value
        self._sparse = None     # {value: count} awaiting a Shared

    def _after_load(self):
        pass

    def _bind(self, shared):
        self._values = shared.values
        if self._sparse is not None:
            words = shared.params.group_init()
            for value, count in self._sparse.items():
                words[1 + shared.index[int(value)]] = int(count)
                words[0] += int(count)
            self.words = words
            self._sparse = None

    def init(self, shared):
        self._sparse = None
        self._values = shared.values
        GroupBase.init(self, shared)

    def load(self, raw):                       # dpd.pyx:141-148
        self._sparse = {int(v): int(c) for v, c in raw['counts'].items()}
        self.words = None

    def dump(self):                            # dpd.pyx:150-157
        if self._sparse is not None:
            return {'counts': dict(self._sparse)}
        counts = self.words[1:].astype(np.int64)
        return {'counts': {int(self._values[i]): int(counts[i])
                           for i in np.flatnonzero(counts)}}

    def protobuf_load(self, message):          # dpd.hpp:161-169
        self.load({'counts': dict(zip(message.keys, message.values))})

    def protobuf_dump(self, message):          # dpd.hpp:171-180
        message.Clear()
        for value, count in sorted(self.dump()['counts'].items()):
            message.keys.append(value)
            message.values.append(count)

    @staticmethod
    def _word(shared, value):
        return shared.remap(value)

    def add_value(self, shared, value):
        self._bind(shared)
        shared.params.group_add_value(self.words, shared.remap(value))

    def remove_value(self, shared, value):
        self._bind(shared)
        shared.params.group_remove_value(self.words, shared.remap(value))

    def score_value(self, shared, value):
        self._bind(shared)
        return shared.params.group_score_value(self.words,
                                               shared.remap(value))

    def score_data(self, shared):
        self._bind(shared)
        return GroupBase.score_data(self, shared)

    def merge(self, shared, source):           # sparse.hpp:163-168 (as built)
        self._bind(shared)
        source._bind(shared)
        self.words += source.words


class Mixture(MixtureBase):
    GROUP = Group

    def append(self, group):
        if group.words is None and self._core is None:
            self._pending.append(group)        # bound at the first init()
        else:
            MixtureBase.append(self, group)

    def _handle(self, shared):
        for i, item in enumerate(self._pending):
            if isinstance(item, Group):
                item._bind(shared)
                self._pending[i] = np.array(item.words, np.uint32)
        return MixtureBase._handle(self, shared)

    _values_of_shared = None

    def __getitem__(self, groupid):
        if self._core is None and isinstance(self._pending[groupid], Group):
            return self._pending[groupid]
        group = MixtureBase.__getitem__(self, groupid)
        group._values = self._values_of_shared
        return group

    def init(self, shared):
        self._values_of_shared = shared.values
        MixtureBase.init(self, shared)

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
