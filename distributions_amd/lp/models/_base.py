"""Shared/Group/Mixture of one component model over the C ABI.

The reference generates one Cython module per model from the same template
(distributions/lp/models/_dd.pyx:31-151; "_X.pyx identical across models").
Here the template is this module; the per-model modules supply the
hyper-parameter and statistics field names.

    Shared   holds hyper-parameters                 (_dd.pyx:31-36, dd.pyx:52-82)
    Group    one group's sufficient statistics on the host, as the 32-bit
             words of dist_group_words()            (_dd.pyx:39-69, dd.pyx:85-113)
    Mixture  the groups' statistics and score caches in HBM
             (dist_mixture_t)                       (_dd.pyx:84-135)
"""
import numpy as np

from ... import _core
from ..random import get_rng  # noqa: F401  (same import surface as the reference)


class ProtobufSerializable(object):
    """distributions/mixins.py:61-72"""

    @classmethod
    def to_protobuf(cls, raw, message):
        model = cls()
        model.load(raw)
        model.protobuf_dump(message)

    @classmethod
    def from_protobuf(cls, message):
        model = cls()
        model.protobuf_load(message)
        return model.dump()


class SharedBase(ProtobufSerializable):
    KIND = None

    def __init__(self):
        self._params = None

    # --- to be provided by the model module --------------------------------
    def load(self, raw):
        raise NotImplementedError

    def dump(self):
        raise NotImplementedError

    # --- SharedMixin / SharedIoMixin (distributions/mixins.py:51-60,99-110) -
    def add_value(self, value):
        pass

    def remove_value(self, value):
        pass

    def realize(self):
        pass

    @classmethod
    def from_dict(cls, raw):
        model = cls()
        model.load(raw)
        return model

    def protobuf_load(self, message):
        self.load({name: _message_get(message, name)
                   for name in self.FIELDS})

    def protobuf_dump(self, message):
        message.Clear()
        for name, value in self.dump().items():
            _message_set(message, name, value)

    @property
    def params(self):
        """the dist_shared_t behind this object"""
        if self._params is None:
            raise RuntimeError("Shared has not been loaded")
        return self._params


def _message_get(message, name):
    value = getattr(message, name)
    try:
        return list(value)
    except TypeError:
        return value


def _message_set(message, name, value):
    if isinstance(value, (list, tuple)):
        getattr(message, name).extend(value)
    else:
        setattr(message, name, value)


class GroupBase(ProtobufSerializable):
    """Model::Group; `words` is the statistics image the ABI uses."""
    SHARED = None

    def __init__(self):
        self.words = None

    def init(self, shared):
        self.words = shared.params.group_init()

    def add_value(self, shared, value):
        shared.params.group_add_value(self.words, self._word(shared, value))

    def add_repeated_value(self, shared, value, count):
        for _ in range(int(count)):
            self.add_value(shared, value)

    def remove_value(self, shared, value):
        shared.params.group_remove_value(self.words, self._word(shared, value))

    def score_value(self, shared, value):
        return shared.params.group_score_value(self.words,
                                               self._word(shared, value))

    def score_data(self, shared):
        return shared.params.group_score_data(self.words)

    def sample_value(self, shared):
        """Group::sample_value (dd.hpp:188-199 and the other models'): one
        draw from the posterior predictive through a fresh Sampler.  Host-side
        (a sampler, not part of the accelerated path): numpy distributions
        seeded from the global engine's stream, so `seed()` makes it
        repeatable; not the reference's own variate stream."""
        sampler = SamplerBase()
        sampler.init(shared, self)
        return sampler.eval(shared)

    @staticmethod
    def _word(shared, value):
        return int(_core.value_words(shared.params.kind, [value])[0])

    # --- GroupIoMixin (distributions/mixins.py:83-96) ------------------------
    @classmethod
    def from_values(cls, model, values=[]):
        group = cls()
        group.init(model)
        for value in values:
            group.add_value(model, value)
        return group

    @classmethod
    def from_dict(cls, raw):
        group = cls()
        group.load(raw)
        return group


def _host_generator():
    """a numpy Generator seeded with the next word of the global engine"""
    return np.random.default_rng(get_rng()())


class SamplerBase(object):
    """Model::Sampler (dd.hpp:201-222, bb.hpp:163-183, gp.hpp:175-191,
    nich.hpp:213-231, bnb.hpp:177-192): init draws the component's
    parameters from the posterior given a group, eval draws a value."""

    def __init__(self):
        self._draw = None

    def init(self, shared, group):
        params = shared.params
        kind = params.kind
        words = np.asarray(group.words)
        rng = _host_generator()
        self._rng = rng
        if kind == _core.KIND_DD:
            dim = params.dim
            a = np.asarray(params.alphas[:dim], np.float64) + words[
                1:1 + dim].astype(np.int32)
            ps = rng.dirichlet(a)
            self._draw = lambda: int(rng.choice(dim, p=ps))
        elif kind == _core.KIND_BB:
            a, b = params.p[0], params.p[1]
            heads, tails = int(np.int32(words[0])), int(np.int32(words[1]))
            p = rng.beta(a + heads, b + tails)
            self._draw = lambda: bool(rng.random() < p)
        elif kind == _core.KIND_GP:
            alpha, inv_beta = params.p[0], params.p[1]
            mean = rng.gamma(alpha + int(words[1]),
                             1.0 / (inv_beta + int(words[0])))
            self._draw = lambda: int(rng.poisson(mean))
        elif kind == _core.KIND_BNB:
            alpha, beta, r = params.p[0], params.p[1], int(params.p[2])
            p = rng.beta(alpha + r * int(words[0]), beta + int(words[1]))
            self._draw = lambda: int(rng.negative_binomial(r, p))
        elif kind == _core.KIND_NICH:
            mu, kappa, sigmasq, nu = (float(x) for x in params.p[:4])
            n = float(np.int32(words[0]))
            mean = float(words[1:2].view(np.float32)[0])
            ctv = float(words[2:3].view(np.float32)[0])
            pk = kappa + n
            pmu = (kappa * mu + mean * n) / pk
            pnu = nu + n
            psig = (nu * sigmasq + ctv + n * kappa * (mu - mean) ** 2 / pk) / pnu
            s2 = pnu * psig / rng.chisquare(pnu)
            centre = rng.normal(pmu, np.sqrt(s2 / pk))
            self._draw = lambda: float(rng.normal(centre, np.sqrt(s2)))
        else:
            raise NotImplementedError(
                "no Sampler for this model (DirichletProcessDiscrete's draws "
                "new values: SURVEY 2.2, out of scope)")

    def eval(self, shared):
        assert self._draw is not None, "Sampler.init first"
        return self._draw()


def sample_group_with(group_cls, shared, size):
    """module.sample_group (lp/models/_dd.pyx:141-151): `size` values from
    ONE component drawn from the prior"""
    group = group_cls()
    group.init(shared)
    sampler = SamplerBase()
    sampler.init(shared, group)
    return [sampler.eval(shared) for _ in range(int(size))]


class MixtureBase(object):
    """Model::Mixture = MixtureSlave<Model, ...> (mixture.hpp:340-450)."""
    GROUP = None

    def __init__(self):
        self._core = None
        self._pending = []     # groups appended before the first init()
        self._key = None

    def _handle(self, shared):
        key = (shared.params.kind, tuple(shared.params.p),
               tuple(shared.params.alphas), shared.params.dim)
        if self._core is None or key != self._key:
            groups = self._pending if self._core is None else [
                self._core.get_group(i) for i in range(len(self._core))]
            self._core = _core.SlaveMixture(shared.params)
            for words in groups:
                self._core.append(np.ascontiguousarray(words, np.uint32))
            self._pending = []
            self._key = key
        return self._core

    def __len__(self):
        return len(self._pending) if self._core is None else len(self._core)

    def __getitem__(self, groupid):
        assert groupid < len(self), "groupid out of bounds"
        group = self.GROUP()
        if self._core is None:
            group.words = self._pending[groupid].copy()
        else:
            group.words = self._core.get_group(groupid)
        group._after_load()
        return group

    def append(self, group):
        if self._core is None:
            self._pending.append(np.array(group.words, np.uint32))
        else:
            self._core.append(np.ascontiguousarray(group.words, np.uint32))

    def clear(self):
        self._pending = []
        if self._core is not None:
            self._core.clear()

    def init(self, shared):
        self._handle(shared).init()

    def add_group(self, shared):
        self._handle(shared).add_group()

    def remove_group(self, shared, groupid):
        self._handle(shared).remove_group(groupid)

    def add_value(self, shared, groupid, value):
        self._handle(shared).add_value(groupid,
                                       GroupBase._word(shared, value))

    def remove_value(self, shared, groupid, value):
        self._handle(shared).remove_value(groupid,
                                          GroupBase._word(shared, value))

    def score_value_group(self, shared, groupid, value):
        return self._handle(shared).score_value_group(
            groupid, GroupBase._word(shared, value))

    def score_value(self, shared, value, scores_accum):
        assert len(scores_accum) == len(self), "scores_accum != len(mixture)"
        assert scores_accum.dtype == np.float32
        self._handle(shared).score_value(GroupBase._word(shared, value),
                                         scores_accum)

    def score_values(self, shared, values, scores_accum):
        """score_value for a batch of values in one launch (extension of
        mixture.hpp:416-425): scores_accum[r, k] accumulates the score of
        values[r] in group k; float32 [len(values), len(mixture)]"""
        assert scores_accum.shape == (len(values), len(self))
        assert scores_accum.dtype == np.float32
        words = [self.GROUP._word(shared, value) for value in values]
        self._handle(shared).score_values(words, scores_accum)

    def validate(self, shared):
        """mixture.hpp:440-444"""
        self._handle(shared).validate()

    def score_data(self, shared):
        return self._handle(shared).score_data()

    def score_data_grid(self, shareds):
        """one score_data per candidate Shared (mixture.hpp:433-438; C++-only
        in the reference, doc/overview.rst:143) -> float32 array"""
        if not len(shareds):
            return np.zeros(0, np.float32)
        return self._handle(shareds[0]).score_data_grid(
            [s.params for s in shareds])
