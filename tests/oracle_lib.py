"""ctypes bindings of the CPU oracle (oracle/liboracle.so) and of the
buildable subset of the real reference (oracle/_ref/libref.so).

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

DD, BB, GP, NICH, DPD, BNB = 0, 1, 2, 3, 4, 5

c_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
c_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
c_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


class Shared(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int),
        ("dim", ctypes.c_int),
        ("p", ctypes.c_float * 4),
        ("alphas", ctypes.c_float * 256),
        ("betas", ctypes.POINTER(ctypes.c_float)),
    ]


def make_shared(kind, **kw):
    s = Shared()
    s.kind = kind
    if kind == DD:
        alphas = kw["alphas"]
        s.dim = len(alphas)
        for i, a in enumerate(alphas):
            s.alphas[i] = a
    elif kind == BB:
        s.p[0], s.p[1] = kw["alpha"], kw["beta"]
    elif kind == GP:
        s.p[0], s.p[1] = kw["alpha"], kw["inv_beta"]
    elif kind == NICH:
        s.p[0], s.p[1], s.p[2], s.p[3] = (kw["mu"], kw["kappa"],
                                          kw["sigmasq"], kw["nu"])
    elif kind == BNB:
        s.p[0], s.p[1], s.p[2] = kw["alpha"], kw["beta"], float(int(kw["r"]))
    elif kind == DPD:
        betas = np.ascontiguousarray(kw["betas"], np.float32)
        s._keep = betas
        s.dim = len(betas)
        s.p[0], s.p[1] = kw["alpha"], kw.get("beta0", 0.0)
        s.betas = betas.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    else:
        raise ValueError(kind)
    return s


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


_ORACLE = None
_REF = None


def oracle():
    global _ORACLE
    if _ORACLE is not None:
        return _ORACLE
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build()
    _ORACLE = bind_oracle(path)
    return _ORACLE


def bind_oracle(path):
    """ctypes signatures of one build of oracle.c (liboracle.so, or the
    -march=native build bench.py's cpu_baseline reports beside it)."""
    L = ctypes.CDLL(path)
    sz = ctypes.c_size_t
    u32 = ctypes.c_uint32
    u64 = ctypes.c_uint64
    f32 = ctypes.c_float
    vp = ctypes.c_void_p
    ci = ctypes.c_int
    u32ptr = ctypes.POINTER(u32)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    for name in ["fast_log", "fast_exp", "fast_lgamma", "fast_lgamma_nu"]:
        sig("orc_" + name, f32, f32)
        sig("orc_vec_" + name, None, sz, c_f32p, c_f32p)
    sig("orc_fast_log_factorial", f32, u32)
    sig("orc_vec_fast_log_factorial", None, sz, c_u32p, c_f32p)
    sig("orc_formula_log_table", None, c_f32p)
    sig("orc_formula_exp_table", None, c_u32p)
    sig("orc_vector_add_subtract", None, sz, c_f32p, c_f32p, c_f32p)
    sig("orc_vector_add_subtract_scalar", None, sz, c_f32p, f32, c_f32p)
    sig("orc_vector_add", None, sz, c_f32p, c_f32p)
    sig("orc_vector_max", f32, sz, c_f32p)
    sig("orc_rng_seed", u32, u64)
    sig("orc_rng_next", u32, u32ptr)
    sig("orc_sample_unif01", f32, u32ptr)
    sig("orc_rng_jump", u32, u32, u64)
    sig("orc_scores_to_likelihoods", f32, sz, c_f32p)
    sig("orc_sample_from_likelihoods", sz, u32ptr, sz, c_f32p, f32)
    sig("orc_sample_from_scores_overwrite", sz, u32ptr, sz, c_f32p)
    sig("orc_sample_from_scores_u", sz, sz, c_f32p, f32)
    sig("orc_log_sum_exp", f32, sz, c_f32p)
    sig("orc_sample_discrete", sz, u32ptr, sz, c_f32p)
    sig("orc_py_score_add_value", f32, f32, f32, ci, ci, ci, ci)
    sig("orc_py_score_remove_value", f32, f32, f32, ci, ci, ci, ci)
    sig("orc_mix_create", vp, f32, f32, ci, ctypes.POINTER(Shared))
    sig("orc_mix_destroy", None, vp)
    sig("orc_mix_driver_init", None, vp, c_i32p, ci)
    sig("orc_mix_driver_add_value", ci, vp, ci)
    sig("orc_mix_driver_remove_value", ci, vp, ci)
    sig("orc_mix_driver_score_value", None, vp, c_f32p)
    sig("orc_mix_size", ci, vp)
    sig("orc_mix_sample_size", ci, vp)
    sig("orc_mix_empty_count", ci, vp)
    sig("orc_mix_get_counts", None, vp, c_i32p)
    sig("orc_mix_get_shifted", None, vp, c_f32p)
    sig("orc_mix_slave_clear", None, vp, ci)
    sig("orc_mix_slave_append_empty", None, vp, ci)
    sig("orc_mix_slave_group_add_value", None, vp, ci, ci, u32)
    sig("orc_mix_slave_init", None, vp, ci)
    sig("orc_mix_slave_add_group", None, vp, ci)
    sig("orc_mix_slave_remove_group", None, vp, ci, ci)
    sig("orc_mix_slave_add_value", None, vp, ci, ci, u32)
    sig("orc_mix_slave_remove_value", None, vp, ci, ci, u32)
    sig("orc_mix_slave_score_value_group", f32, vp, ci, ci, u32)
    sig("orc_mix_slave_score_value", None, vp, ci, u32, c_f32p)
    sig("orc_mix_slave_size", ci, vp, ci)
    sig("orc_mix_slave_get_group", None, vp, ci, ci, c_u32p)
    sig("orc_group_score_value", f32, ctypes.POINTER(Shared), c_u32p, u32)
    sig("orc_group_score_data", f32, ctypes.POINTER(Shared), c_u32p)
    sig("orc_mix_slave_score_data", f32, vp, ci)
    sig("orc_py_score_counts", f32, f32, f32, c_i32p, sz)
    sig("orc_py_sample_assignments", None, f32, f32, ci, u32ptr, c_i32p)
    sig("orc_mix_tracker_init", None, vp, ci)
    sig("orc_mix_tracker_add_group", None, vp)
    sig("orc_mix_tracker_remove_group", None, vp, u32)
    sig("orc_mix_packed_to_global", u32, vp, u32)
    sig("orc_mix_global_size", u32, vp)
    sig("orc_mix_global_to_packed", u32, vp, u32)
    pp = ctypes.POINTER(ctypes.c_void_p)
    sig("orc_mix_init_from_assignments", None, vp, sz, pp, c_u32p, ci, ci,
        c_u32p)
    sig("orc_mix_gibbs_sequential", None, vp, sz, sz, pp, c_u32p, u32ptr)
    sig("orc_mix_init_sequential", None, vp, sz, sz, pp, c_u32p, u32ptr, ci)
    sig("orc_mix_load_state", None, vp, ci, c_i32p, pp, c_u32p, u32)
    sig("orc_mixture_benchmark_loop", f32, vp, sz, c_u32p, c_u32p, sz)
    sig("orc_mix_gibbs_batch", None, vp, sz, sz, pp, c_u32p, u32, u64)
    sig("orc_mix_batch_row_scores", ci, vp, c_u32p, u32, c_f32p)
    return L


def ref():
    """The real reference subset; None when it was never built."""
    global _REF
    if _REF is not None:
        return _REF
    path = os.path.join(ORACLE_DIR, "_ref", "libref.so")
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/src"):
            build()
        else:
            return None
    L = ctypes.CDLL(path)
    sz = ctypes.c_size_t
    for name in ["fast_log", "fast_exp", "fast_lgamma", "fast_lgamma_nu"]:
        fn = getattr(L, "ref_" + name)
        fn.argtypes = [sz, c_f32p, c_f32p]
        fn.restype = None
    L.ref_fast_log_factorial.argtypes = [sz, c_u32p, c_f32p]
    L.ref_vector_add_subtract.argtypes = [sz, c_f32p, c_f32p, c_f32p]
    L.ref_vector_add_subtract_scalar.argtypes = [sz, c_f32p, ctypes.c_float,
                                                 c_f32p]
    L.ref_vector_add.argtypes = [sz, c_f32p, c_f32p]
    L.ref_vector_max.argtypes = [sz, c_f32p]
    L.ref_vector_max.restype = ctypes.c_float
    for name in ["log", "exp", "lgamma"]:
        getattr(L, "ref_vector_" + name).argtypes = [sz, c_f32p]
    L.ref_log_table.argtypes = [c_f32p]
    vp = ctypes.c_void_p
    L.ref_driver_new.restype = vp
    L.ref_driver_delete.argtypes = [vp]
    L.ref_driver_init.argtypes = [vp, c_i32p, sz]
    L.ref_driver_add_value.argtypes = [vp, sz]
    L.ref_driver_remove_value.argtypes = [vp, sz]
    L.ref_driver_size.argtypes = [vp]
    L.ref_driver_size.restype = sz
    L.ref_driver_sample_size.argtypes = [vp]
    L.ref_driver_sample_size.restype = sz
    L.ref_driver_counts.argtypes = [vp, c_i32p]
    L.ref_driver_empty_count.argtypes = [vp]
    L.ref_driver_empty_count.restype = sz
    L.ref_driver_is_empty.argtypes = [vp, sz]
    L.ref_tracker_new.restype = vp
    L.ref_tracker_delete.argtypes = [vp]
    L.ref_tracker_init.argtypes = [vp, sz]
    L.ref_tracker_add_group.argtypes = [vp]
    L.ref_tracker_remove_group.argtypes = [vp, ctypes.c_uint32]
    for name in ["packed_to_global", "global_to_packed"]:
        fn = getattr(L, "ref_tracker_" + name)
        fn.argtypes = [vp, ctypes.c_uint32]
        fn.restype = ctypes.c_uint32
    L.ref_tracker_packed_size.argtypes = [vp]
    L.ref_tracker_packed_size.restype = sz
    _REF = L
    return L


def value_words(kind, values):
    """Values of one feature as the 32-bit words the oracle takes."""
    if kind == NICH:
        return np.ascontiguousarray(values, np.float32).view(np.uint32)
    return np.ascontiguousarray(values).astype(np.uint32)


def ptr_array(arrays):
    arr = (ctypes.c_void_p * len(arrays))()
    for i, a in enumerate(arrays):
        arr[i] = a.ctypes.data
    return arr


class OracleMixture(object):
    """PY driver + feature slaves + id tracker, in the oracle."""

    def __init__(self, alpha, d, shareds, lib=None):
        self.L = lib if lib is not None else oracle()
        self.shareds = list(shareds)
        arr = (Shared * max(1, len(shareds)))(*shareds)
        self._arr = arr
        self.h = self.L.orc_mix_create(alpha, d, len(shareds), arr)
        self.F = len(shareds)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_mix_destroy(self.h)
            self.h = None

    def __len__(self):
        return self.L.orc_mix_size(self.h)

    def counts(self):
        out = np.zeros(len(self), np.int32)
        self.L.orc_mix_get_counts(self.h, out)
        return out

    def init_from_assignments(self, values, assign_packed, nonempty, empty):
        self.values = [value_words(s.kind, v)
                       for s, v in zip(self.shareds, values)]
        self.n_rows = len(assign_packed)
        self._vals = ptr_array(self.values)
        out = np.zeros(self.n_rows, np.uint32)
        self.L.orc_mix_init_from_assignments(
            self.h, self.n_rows, self._vals,
            np.ascontiguousarray(assign_packed, np.uint32), nonempty, empty,
            out)
        self.assign = out
        return out

    # the accessors adopt() reads from an engine, so that one oracle can also
    # adopt another's state (tests of load_state itself)
    @property
    def core(self):
        return self

    def packed_to_global(self, k):
        return self.L.orc_mix_packed_to_global(self.h, k)

    def global_size(self):
        return self.L.orc_mix_global_size(self.h)

    def assignments(self):
        return self.assign

    def adopt(self, gpu, values):
        """Take over the engine's current state (group order, sizes,
        statistics, id maps, assignments): the oracle then follows the GPU
        from a state only the GPU has reached (e.g. sweep 2 at N = 10M)."""
        K = len(gpu)
        counts = np.ascontiguousarray(gpu.counts(), np.int32)
        blocks = [np.ascontiguousarray(np.concatenate(
            [gpu.get_group(f, g) for g in range(K)]).astype(np.uint32))
            for f in range(self.F)]
        p2g = np.array([gpu.core.packed_to_global(k) for k in range(K)],
                       np.uint32)
        self.values = [value_words(s.kind, v)
                       for s, v in zip(self.shareds, values)]
        self.n_rows = len(self.values[0]) if self.values else 0
        self._vals = ptr_array(self.values)
        self._blocks = blocks
        self.L.orc_mix_load_state(self.h, K, counts, ptr_array(blocks), p2g,
                                  int(gpu.core.global_size()))
        self.assign = np.ascontiguousarray(gpu.assignments(), np.uint32).copy()

    def gibbs_sequential(self, row_begin, row_end, rng_state):
        st = ctypes.c_uint32(rng_state)
        self.L.orc_mix_gibbs_sequential(self.h, row_begin, row_end,
                                        self._vals, self.assign,
                                        ctypes.byref(st))
        return st.value

    def init_empty(self, values, empty=1):
        """no rows assigned yet: `empty` empty groups (mixture.init(model) of
        a fresh mixture, examples/mixture/main.py:222-224)"""
        n = len(values[0])
        self.init_from_assignments([v[:0] for v in values],
                                   np.zeros(0, np.uint32), 0, empty)
        self.values = [value_words(s.kind, v)
                       for s, v in zip(self.shareds, values)]
        self.n_rows = n
        self._vals = ptr_array(self.values)
        self.assign = np.full(n, 0xFFFFFFFF, np.uint32)

    def init_sequential(self, row_begin, row_end, rng_state, prior_only=False):
        st = ctypes.c_uint32(rng_state)
        self.L.orc_mix_init_sequential(self.h, row_begin, row_end, self._vals,
                                       self.assign, ctypes.byref(st),
                                       1 if prior_only else 0)
        return st.value

    def gibbs_batch(self, row_begin, row_end, seed_state, draw_base):
        self.L.orc_mix_gibbs_batch(self.h, row_begin, row_end, self._vals,
                                   self.assign, seed_state, draw_base)

    def row_scores(self, row, packed_group):
        x = np.array([v[row] for v in self.values], np.uint32)
        out = np.zeros(len(self) + 1, np.float32)
        kl = self.L.orc_mix_batch_row_scores(self.h, x, packed_group, out)
        return out[:kl]

    def get_group(self, f, g):
        s = self.shareds[f]
        n = 1 + s.dim if s.kind in (DD, DPD) else (
            2 if s.kind in (BB, BNB) else 3)
        out = np.zeros(n, np.uint32)
        self.L.orc_mix_slave_get_group(self.h, f, g, out)
        return out


def _phase_sigs(L):
    sz = ctypes.c_size_t
    vp = ctypes.c_void_p
    pp = ctypes.POINTER(ctypes.c_void_p)
    L.orc_mix_batch_sample.restype = None
    L.orc_mix_batch_sample.argtypes = [vp, sz, sz, pp, c_u32p, ctypes.c_uint32,
                                       ctypes.c_uint64, ctypes.c_uint64,
                                       c_u32p, c_u32p]
    L.orc_mix_apply_moves.restype = None
    L.orc_mix_apply_moves.argtypes = [vp, sz, sz, pp, c_u32p, c_u32p, c_u32p]
    L.orc_mix_apply_moves_part.restype = None
    L.orc_mix_apply_moves_part.argtypes = [vp, sz, sz, pp, c_u32p, c_u32p,
                                           c_u32p, ctypes.c_int]
    L.orc_mix_replay_ordered.restype = None
    L.orc_mix_replay_ordered.argtypes = [vp, sz, vp, vp, vp, ctypes.c_int]
    L.orc_mix_stat_words.restype = sz
    L.orc_mix_stat_words.argtypes = [vp]
    L.orc_mix_export_stats.restype = None
    L.orc_mix_export_stats.argtypes = [vp, c_i32p]
    L.orc_mix_import_stats.restype = None
    L.orc_mix_import_stats.argtypes = [vp, c_i32p]
    L.orc_mix_batch_finish.restype = None
    L.orc_mix_batch_finish.argtypes = [vp, c_i32p]


class OracleBackend(object):
    """CPU stand-in with the interface distributions_amd.engine.ShardedGibbs
    drives (GibbsEngine's batch_* / *_stats_dev methods), built on the
    oracle.  TESTS ONLY: lets the multi-rank driver run under gloo without a
    GPU.  "device pointers" are addresses of CPU torch tensors here."""

    def __init__(self, mix, row_offset):
        self.m = mix
        self.L = mix.L
        _phase_sigs(self.L)
        self.row_offset = row_offset
        self._open = None

    @staticmethod
    def _view(ptr, n):
        buf = (ctypes.c_int32 * n).from_address(ptr)
        return np.frombuffer(buf, dtype=np.int32)

    def stat_words(self):
        return self.L.orc_mix_stat_words(self.m.h)

    def export_stats_dev(self, ptr):
        w = np.zeros(self.stat_words(), np.int32)
        self.L.orc_mix_export_stats(self.m.h, w)
        self._view(ptr, w.size)[:] = w

    def import_stats_dev(self, ptr):
        w = np.ascontiguousarray(self._view(ptr, self.stat_words()).copy())
        self.L.orc_mix_import_stats(self.m.h, w)
        self.L.orc_mix_batch_finish(self.m.h, self.m.counts())

    def batch_sample(self, r0, r1, seed_state, draw_base=0):
        n = r1 - r0
        old = np.zeros(n + 1, np.uint32)
        new = np.zeros(n + 1, np.uint32)
        self.L.orc_mix_batch_sample(self.m.h, r0, r1, self.m._vals,
                                    self.m.assign, seed_state, draw_base,
                                    self.row_offset, old, new)
        self._open = (r0, r1, old, new, self.m.counts().copy())

    def _snapshot(self):
        w = np.zeros(self.stat_words(), np.int32)
        self.L.orc_mix_export_stats(self.m.h, w)
        return w

    def batch_apply_local(self):
        r0, r1, old, new, _ = self._open
        self.L.orc_mix_apply_moves(self.m.h, r0, r1, self.m._vals,
                                   self.m.assign, old, new)

    def batch_delta_dev(self, ptr):
        before = self._snapshot()
        r0, r1, old, new, _ = self._open
        self.L.orc_mix_apply_moves_part(self.m.h, r0, r1, self.m._vals,
                                        self.m.assign, old, new, 1)
        after = self._snapshot()
        self.L.orc_mix_import_stats(self.m.h, before)   # undo: delta only
        self._view(ptr, before.size)[:] = after - before

    def feature_is_ordered(self, f):
        return self.m.shareds[f].kind in (GP, NICH)

    def ordered_features(self):
        return sum(self.feature_is_ordered(f)
                   for f in range(len(self.m.shareds)))

    def batch_moves_dev(self, old_ptr, new_ptr):
        r0, r1, old, new, _ = self._open
        n = r1 - r0
        self._view(old_ptr, n)[:] = old[:n].view(np.int32)
        self._view(new_ptr, n)[:] = new[:n].view(np.int32)

    def replay_ordered_dev(self, old_ptr, new_ptr, value_ptrs, n, reset):
        F = len(value_ptrs)
        keep = []
        arr = (ctypes.POINTER(ctypes.c_uint32) * F)()
        for f, ptr in enumerate(value_ptrs):
            if ptr:
                v = np.ascontiguousarray(self._view(ptr, n).view(np.uint32))
                keep.append(v)
                arr[f] = v.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        new = np.ascontiguousarray(self._view(new_ptr, n).view(np.uint32))
        old = (np.ascontiguousarray(self._view(old_ptr, n).view(np.uint32))
               if old_ptr else None)
        self.L.orc_mix_replay_ordered(
            self.m.h, n, ctypes.cast(arr, ctypes.c_void_p),
            old.ctypes.data if old is not None else None,
            new.ctypes.data, int(reset))
        if reset:
            self.L.orc_mix_batch_finish(self.m.h, self.m.counts())

    def batch_apply_delta_dev(self, ptr):
        cur = self._snapshot()
        cur += self._view(ptr, cur.size)
        self.L.orc_mix_import_stats(self.m.h, np.ascontiguousarray(cur))

    def batch_finish(self):
        snap = np.ascontiguousarray(self._open[4], np.int32)
        self.L.orc_mix_batch_finish(self.m.h, snap)
        self._open = None
