"""Mirror of distributions/lp/mixture.pyx: MixtureIdTracker
(include/distributions/mixture.hpp:460-521)."""
from .. import _core


class MixtureIdTracker(object):
    def __init__(self):
        self._core = _core.IdTracker()

    def init(self, group_count=0):
        self._core.init(group_count)

    def add_group(self):
        self._core.add_group()

    def remove_group(self, packed):
        self._core.remove_group(packed)

    def packed_to_global(self, packed):
        return self._core.packed_to_global(packed)

    def global_to_packed(self, global_):
        return self._core.global_to_packed(global_)
