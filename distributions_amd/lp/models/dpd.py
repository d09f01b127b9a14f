"""DirichletProcessDiscrete -- mirror of distributions/lp/models/dpd.pyx.

Values are kept in a dense remap: every value the Shared knows owns a dense
slot 0..V-1 for life (dist_dpd_shared_t); OTHER is 0xFFFFFFFF (dpd.hpp:56).
Shared.add_value / remove_value / realize are the reference's stick-breaking
(dpd.hpp:66-101): a new value appends a slot (or takes a freed one), groups
and mixtures built before it are widened when they next meet the Shared.
"""
import numpy as np

from ... import _core
from ._base import SharedBase, GroupBase, MixtureBase, get_rng

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
    """dpd.hpp:59-153 through dist_dpd_shared_t: the stick-breaking state
    (which values exist, their betas and row counts, beta0) lives in the
    library; `params` is the dense view the groups and mixtures take."""

    def __init__(self):
        SharedBase.__init__(self)
        self._core = _core.DpdShared()
        self._seen = None

    # --- the dense view -----------------------------------------------------
    @property
    def params(self):
        if self._seen != self._core.version:
            self._params = self._core.params()
            slot_values, _, _ = self._core.dump()
            self._values = [None if v == OTHER else int(v)
                            for v in slot_values]
            self._seen = self._core.version
        return self._params

    @property
    def version(self):
        return self._core.version

    @property
    def values(self):
        """dense slot -> value (None: a slot whose value is gone)"""
        self.params
        return self._values

    @property
    def index(self):
        return {v: i for i, v in enumerate(self.values) if v is not None}

    @property
    def gamma(self):
        return self._core.scalars()[0]

    @property
    def alpha(self):
        return self._core.scalars()[1]

    @property
    def beta0(self):
        return self._core.scalars()[2]

    @property
    def counts(self):
        values, _, counts = self._core.dump()
        return {int(v): int(c) for v, c in zip(values, counts) if v != OTHER}

    def remap(self, value):
        return self._core.slot(int(value))

    # --- Shared's own interface (dpd.pyx:69-134, _dpd.pyx:31-45) ------------
    def load(self, raw):
        betas = {int(v): float(b) for v, b in raw['betas'].items()}
        counts = {int(v): int(c) for v, c in raw.get('counts', {}).items()}
        values = sorted(betas)
        self._core.load(float(raw.get('gamma', 1.0)), float(raw['alpha']),
                        values, [betas[v] for v in values],
                        [counts.get(v, 0) for v in values])

    def dump(self):
        values, betas, counts = self._core.dump()
        live = values != OTHER
        return {'gamma': self.gamma, 'alpha': self.alpha,
                'betas': {int(v): float(b)
                          for v, b in zip(values[live], betas[live])},
                'counts': {int(v): int(c)
                           for v, c in zip(values[live], counts[live])}}

    def add_value(self, value):
        """dpd.hpp:66-74: the first row of a new value breaks a piece off the
        stick, with the global engine (lp/models/_dpd.pyx:38-39)"""
        rng = get_rng()
        rng.state = self._core.add_value(int(value), rng.state)

    def remove_value(self, value):
        """dpd.hpp:76-83"""
        self._core.remove_value(int(value))

    def realize(self):
        """dpd.hpp:85-101"""
        rng = get_rng()
        rng.state = self._core.realize(rng.state)

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
        raw = self.dump()
        for value in sorted(raw['betas']):
            message.values.append(value)
            message.betas.append(raw['betas'][value])
            message.counts.append(raw['counts'][value])


def _widen(words, shared):
    """a group's counts in the Shared's CURRENT dense layout: values that
    appeared after the group was built get a zero (dpd.hpp:66-74 grows the
    Shared, never re-orders it)"""
    need = 1 + shared.params.dim
    if len(words) >= need:
        return words
    out = np.zeros(need, np.uint32)
    out[:len(words)] = words
    return out


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
                words[1 + shared.remap(value)] = int(count)
                words[0] += int(count)
            self.words = words
            self._sparse = None
        elif self.words is not None:
            self.words = _widen(self.words, shared)

    def init(self, shared):
        self._sparse = None
        self._values = shared.values
        if shared.params.dim == 0:      # (a Shared without a value yet)
            self.words = np.zeros(1, np.uint32)
            return
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
        slot = shared.remap(value)
        if shared.params.dim == 0:      # (only OTHER exists: dpd.hpp:223-232)
            alpha = np.float32(shared.alpha)
            numer = np.float32(alpha * np.float32(shared.beta0))
            return float(_core.vector_log(np.array([numer / alpha],
                                                   np.float32))[0])
        return shared.params.group_score_value(self.words, slot)

    def score_data(self, shared):
        self._bind(shared)
        if shared.params.dim == 0:
            return 0.0
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
        """the device mixture of the Shared's current dense view: rebuilt
        (counts carried over, caches re-initialised) when the Shared gained
        or lost a value since -- the reference creates and destroys
        per-value score vectors as values come and go (dpd.hpp:430-469)"""
        for i, item in enumerate(self._pending):
            if isinstance(item, Group):
                item._bind(shared)
                self._pending[i] = np.array(item.words, np.uint32)
        if shared.params.dim == 0:
            # a Shared without a value yet (the reference's EXAMPLES start
            # there, dpd.pyx:52-66): nothing to count under, the groups -- all
            # empty -- stay on the host until the first value exists
            if self._core is not None:
                self._pending = [self._core.get_group(i)[:1].copy()
                                 for i in range(len(self._core))]
                self._core = None
                self._inited_empty = True
            return None
        key = (id(shared), shared.version)
        if self._core is None or key != self._key:
            fresh = self._core is None and not self._inited_empty
            groups = self._pending if self._core is None else [
                self._core.get_group(i) for i in range(len(self._core))]
            self._core = _core.SlaveMixture(shared.params)
            for words in groups:
                self._core.append(_widen(
                    np.ascontiguousarray(words, np.uint32), shared))
            if not fresh:
                self._core.init()
            self._pending = []
            self._inited_empty = False
            self._key = key
            self._values_of_shared = shared.values
        return self._core

    _values_of_shared = None
    _inited_empty = False

    def _other_score(self, shared):
        """a group without rows scores OTHER, the only value there is, with
        fast_log(alpha * beta0 / alpha) (dpd.hpp:223-232)"""
        alpha = np.float32(shared.alpha)
        numer = np.float32(alpha * np.float32(shared.beta0))
        return float(_core.vector_log(np.array([numer / alpha],
                                               np.float32))[0])

    def __getitem__(self, groupid):
        if self._core is None and isinstance(self._pending[groupid], Group):
            return self._pending[groupid]
        group = MixtureBase.__getitem__(self, groupid)
        group._values = self._values_of_shared
        return group

    def init(self, shared):
        self._values_of_shared = shared.values
        if self._handle(shared) is None:
            self._inited_empty = True
            return
        MixtureBase.init(self, shared)

    def add_group(self, shared):
        if self._handle(shared) is None:
            self._pending.append(np.zeros(1, np.uint32))
            return
        MixtureBase.add_group(self, shared)

    def remove_group(self, shared, groupid):
        if self._handle(shared) is None:      # (packed_remove, vector.hpp:47-56)
            assert groupid < len(self._pending), "groupid out of bounds"
            self._pending[groupid] = self._pending[-1]
            self._pending.pop()
            return
        MixtureBase.remove_group(self, shared, groupid)

    def add_value(self, shared, groupid, value):
        self._handle(shared).add_value(groupid, shared.remap(value))

    def remove_value(self, shared, groupid, value):
        self._handle(shared).remove_value(groupid, shared.remap(value))

    def score_value_group(self, shared, groupid, value):
        slot = shared.remap(value)
        if self._handle(shared) is None:
            assert groupid < len(self), "groupid out of bounds"
            return self._other_score(shared)
        return self._handle(shared).score_value_group(groupid, slot)

    def score_value(self, shared, value, scores_accum):
        assert len(scores_accum) == len(self), "scores_accum != len(mixture)"
        slot = shared.remap(value)
        if self._handle(shared) is None:
            scores_accum += np.float32(self._other_score(shared))
            return
        self._handle(shared).score_value(slot, scores_accum)

    def score_values(self, shared, values, scores_accum):
        if self._handle(shared) is None:
            for value in values:
                shared.remap(value)
            scores_accum += np.float32(self._other_score(shared))
            return
        MixtureBase.score_values(self, shared, values, scores_accum)

    def score_data(self, shared):
        if self._handle(shared) is None:
            return 0.0      # (no rows: dpd.hpp:234-250 sums nothing)
        return MixtureBase.score_data(self, shared)

    def validate(self, shared):
        if self._handle(shared) is not None:
            MixtureBase.validate(self, shared)
