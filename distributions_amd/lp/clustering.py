"""Mirror of distributions/lp/clustering.pyx: PitmanYor(alpha, d) (CRP when
d = 0) with PitmanYor.Mixture, LowEntropy(dataset_size) with
LowEntropy.Mixture, count_assignments."""
import numpy as np

from .. import _core


def count_assignments(assignments):
    """Clustering<int>::count_assignments (src/clustering.cc:37-63):
    {value_id: group_id} -> group sizes, group ids being 0..G-1"""
    counts = []
    for group_id in assignments.values():
        if group_id >= len(counts):
            counts.extend([0] * (group_id + 1 - len(counts)))
        counts[group_id] += 1
    return counts


class PitmanYorMixture(object):
    """PitmanYor::Mixture = CachedMixture (clustering.hpp:126-234)"""

    def __init__(self):
        self._core = _core.PyMixture()

    def __len__(self):
        return len(self._core)

    @property
    def empty_groupids(self):
        return iter(self._core.empty_groupids())

    def init(self, model, counts):
        self._core.init(model.alpha, model.d, list(counts))

    def add_value(self, model, groupid):
        return self._core.add_value(model.alpha, model.d, groupid)

    def remove_value(self, model, groupid):
        return self._core.remove_value(model.alpha, model.d, groupid)

    def score_value(self, model, scores):
        assert scores.dtype == np.float32
        self._core.score_value(model.alpha, model.d, scores)

    def counts(self):
        return self._core.counts()

    def score_data(self, model):
        return self._core.score_data(model.alpha, model.d)


class PitmanYor(object):
    EXAMPLES = [
        {'alpha': 1., 'd': 0.},
        {'alpha': 1., 'd': 0.1},
        {'alpha': 1., 'd': 0.9},
        {'alpha': 10., 'd': 0.1},
        {'alpha': 0.1, 'd': 0.1},
    ]
    Mixture = PitmanYorMixture

    def __init__(self, **kwargs):                # lp/clustering.pyx:144-154
        if kwargs:
            self.load(kwargs)
        else:
            self.alpha = 1.0
            self.d = 0.0

    def load(self, raw):
        alpha = float(np.float32(raw['alpha']))
        d = float(np.float32(raw['d']))
        assert 0 < alpha
        assert 0 <= d and d < 1
        self.alpha = alpha
        self.d = d

    def dump(self):
        return {'alpha': self.alpha, 'd': self.d}

    @classmethod
    def from_dict(cls, raw):
        model = cls()
        model.load(raw)
        return model

    def protobuf_load(self, message):
        self.load({'alpha': message.alpha, 'd': message.d})

    def protobuf_dump(self, message):
        message.Clear()
        message.alpha = self.alpha
        message.d = self.d

    def score_add_value(self, group_size, nonempty_group_count, sample_size,
                        empty_group_count=1):
        return _core.py_score_add_value(self.alpha, self.d, group_size,
                                        nonempty_group_count, sample_size,
                                        empty_group_count)

    def score_remove_value(self, group_size, nonempty_group_count,
                           sample_size, empty_group_count=1):
        return _core.py_score_remove_value(self.alpha, self.d, group_size,
                                           nonempty_group_count, sample_size,
                                           empty_group_count)

    def sample_assignments(self, size):
        from .random import get_rng
        rng = get_rng()
        out, rng.state = _core.py_sample_assignments(self.alpha, self.d,
                                                     int(size), rng.state)
        return [int(a) for a in out]

    def score_counts(self, counts):
        return _core.py_score_counts(self.alpha, self.d, list(counts))


class LowEntropyMixture(object):
    """LowEntropy::Mixture = MixtureDriver<LowEntropy, int>
    (mixture.hpp:48-163; lp/clustering.pyx:343-380)"""

    def __init__(self):
        self._core = _core.LeMixture()

    def __len__(self):
        return len(self._core)

    @property
    def empty_groupids(self):
        return iter(self._core.empty_groupids())

    def init(self, model, counts):
        self._core.init(list(counts))

    def add_value(self, model, groupid):
        return self._core.add_value(groupid)

    def remove_value(self, model, groupid):
        return self._core.remove_value(groupid)

    def score_value(self, model, scores):
        assert scores.dtype == np.float32
        self._core.score_value(model.dataset_size, scores)

    def counts(self):
        return self._core.counts()

    def score_data(self, model):
        return self._core.score_data(model.dataset_size)


class LowEntropy(object):
    """Clustering<int>::LowEntropy (clustering.hpp:245-331;
    lp/clustering.pyx:277-340)"""
    EXAMPLES = [
        {'dataset_size': 5},
        {'dataset_size': 10},
        {'dataset_size': 100},
        {'dataset_size': 1000},
    ]
    Mixture = LowEntropyMixture

    def __init__(self, **kwargs):
        if kwargs:
            self.load(kwargs)
        else:
            self.dataset_size = 0

    def load(self, raw):
        dataset_size = int(raw['dataset_size'])
        assert dataset_size >= 0
        self.dataset_size = dataset_size

    def dump(self):
        return {'dataset_size': self.dataset_size}

    @classmethod
    def from_dict(cls, raw):
        model = cls()
        model.load(raw)
        return model

    def protobuf_load(self, message):
        self.load({'dataset_size': message.dataset_size})

    def protobuf_dump(self, message):
        message.Clear()
        message.dataset_size = self.dataset_size

    def score_add_value(self, group_size, nonempty_group_count, sample_size,
                        empty_group_count=1):
        return _core.le_score_add_value(self.dataset_size, group_size,
                                        nonempty_group_count, sample_size,
                                        empty_group_count)

    def score_remove_value(self, group_size, nonempty_group_count,
                           sample_size, empty_group_count=1):
        return _core.le_score_remove_value(self.dataset_size, group_size,
                                           nonempty_group_count, sample_size,
                                           empty_group_count)

    def score_counts(self, counts):
        return _core.le_score_counts(self.dataset_size, list(counts))

    def log_partition_function(self, sample_size):
        return _core.le_log_partition_function(int(sample_size))

    def sample_assignments(self, size):
        from .random import get_rng
        rng = get_rng()
        out, rng.state = _core.le_sample_assignments(self.dataset_size,
                                                     int(size), rng.state)
        return [int(a) for a in out]
