# cython: language_level=3, boundscheck=False, wraparound=False
"""Cython binding of libdistributions_hip's C ABI (include/distributions_hip.h).

This is the one extension module of the package; the modules under
distributions_amd/lp re-export its classes under the names of the reference's
distributions.lp wrappers (distributions/lp/models/_dd.pyx etc.).  It does no
arithmetic of its own: every score, cache entry and sample comes from the HIP
library.  Errors of the library surface as RuntimeError, like the reference's
`except +` translation of std::runtime_error.
"""
from libc.stdint cimport uint8_t, uint32_t, uint64_t, int32_t
from libc.stddef cimport size_t
from libc.string cimport memset, memcpy
from libc.stdlib cimport malloc, free

import numpy as np
cimport numpy as cnp

cnp.import_array()

cdef extern from "distributions_hip.h" nogil:
    enum:
        DIST_DD
        DIST_BB
        DIST_GP
        DIST_NICH
        DIST_DPD
        DIST_BNB
    ctypedef struct dist_shared_t:
        int kind
        int dim
        float p[4]
        float alphas[256]
        const float * betas
    size_t dist_group_words(const dist_shared_t *)
    int dist_abi_version()
    const char * dist_last_error()
    int dist_device_count(int *)
    int dist_set_device(int)
    int dist_synchronize()
    int dist_set_stream(void *)
    uint32_t dist_rng_seed(uint64_t)
    uint32_t dist_rng_next(uint32_t *)
    float dist_rng_unif01(uint32_t *)
    uint32_t dist_rng_jump(uint32_t, uint64_t)
    int dist_vector_log(size_t, const float *, float *)
    int dist_vector_exp(size_t, const float *, float *)
    int dist_vector_lgamma(size_t, const float *, float *)
    int dist_vector_lgamma_nu(size_t, const float *, float *)
    int dist_vector_log_factorial(size_t, const uint32_t *, float *)
    int dist_sample_from_scores_overwrite(uint32_t *, size_t, float *, size_t *)
    int dist_scores_to_likelihoods(size_t, float *, float *)
    int dist_sample_from_likelihoods(uint32_t *, size_t, const float *, float,
                                     size_t *)
    int dist_log_sum_exp(size_t, const float *, float *)
    int dist_py_score_add_value(float, float, int, int, int, int, float *)
    int dist_py_score_remove_value(float, float, int, int, int, int, float *)
    int dist_py_score_counts(float, float, const int *, size_t, float *)
    int dist_py_sample_assignments(float, float, int, uint32_t *, int *)

    ctypedef struct dist_py_mixture_t:
        pass
    ctypedef struct dist_le_mixture_t:
        pass
    int dist_le_score_add_value(int, int, int, int, int, float *)
    int dist_le_score_remove_value(int, int, int, int, int, float *)
    int dist_le_log_partition_function(int, float *)
    int dist_le_sample_assignments(int, int, uint32_t *, int *)
    int dist_le_score_counts(int, const int *, size_t, float *)
    dist_le_mixture_t * dist_le_mixture_create()
    void dist_le_mixture_destroy(dist_le_mixture_t *)
    int dist_le_mixture_init(dist_le_mixture_t *, const int *, size_t)
    int dist_le_mixture_add_value(dist_le_mixture_t *, size_t, int *)
    int dist_le_mixture_remove_value(dist_le_mixture_t *, size_t, int *)
    int dist_le_mixture_score_value(const dist_le_mixture_t *, int, float *,
                                    size_t)
    int dist_le_mixture_score_data(const dist_le_mixture_t *, int, float *)
    size_t dist_le_mixture_size(const dist_le_mixture_t *)
    size_t dist_le_mixture_sample_size(const dist_le_mixture_t *)
    int dist_le_mixture_counts(const dist_le_mixture_t *, int *)
    dist_py_mixture_t * dist_py_mixture_create()
    void dist_py_mixture_destroy(dist_py_mixture_t *)
    int dist_py_mixture_init(dist_py_mixture_t *, float, float, const int *,
                             size_t)
    int dist_py_mixture_add_value(dist_py_mixture_t *, float, float, size_t,
                                  int *)
    int dist_py_mixture_remove_value(dist_py_mixture_t *, float, float, size_t,
                                     int *)
    int dist_py_mixture_score_value(const dist_py_mixture_t *, float, float,
                                    float *, size_t)
    int dist_py_mixture_score_data(const dist_py_mixture_t *, float, float,
                                   float *)
    size_t dist_py_mixture_size(const dist_py_mixture_t *)
    size_t dist_py_mixture_sample_size(const dist_py_mixture_t *)
    int dist_py_mixture_counts(const dist_py_mixture_t *, int *)
    size_t dist_py_mixture_empty_groupids(const dist_py_mixture_t *, size_t *,
                                          size_t)

    ctypedef struct dist_mixture_t:
        pass
    dist_mixture_t * dist_mixture_create(const dist_shared_t *)
    void dist_mixture_destroy(dist_mixture_t *)
    int dist_mixture_clear(dist_mixture_t *)
    int dist_mixture_append(dist_mixture_t *, const uint32_t *)
    int dist_mixture_get_group(const dist_mixture_t *, size_t, uint32_t *)
    size_t dist_mixture_size(const dist_mixture_t *)
    int dist_mixture_init(dist_mixture_t *)
    int dist_mixture_add_group(dist_mixture_t *)
    int dist_mixture_remove_group(dist_mixture_t *, size_t)
    int dist_mixture_add_value(dist_mixture_t *, size_t, uint32_t)
    int dist_mixture_remove_value(dist_mixture_t *, size_t, uint32_t)
    int dist_mixture_score_value_group(const dist_mixture_t *, size_t,
                                       uint32_t, float *)
    int dist_mixture_score_value(const dist_mixture_t *, uint32_t, float *,
                                 size_t)
    int dist_mixture_score_data(const dist_mixture_t *, float *)
    int dist_mixture_validate(const dist_mixture_t *)
    int dist_mixture_score_values(const dist_mixture_t *, const uint32_t *,
                                  size_t, float *, size_t)
    int dist_mixture_score_data_grid(const dist_mixture_t *,
                                     const dist_shared_t *, size_t, float *)

    int dist_group_init(const dist_shared_t *, uint32_t *)
    int dist_group_add_value(const dist_shared_t *, uint32_t *, uint32_t)
    int dist_group_remove_value(const dist_shared_t *, uint32_t *, uint32_t)
    int dist_group_score_value(const dist_shared_t *, const uint32_t *,
                               uint32_t, float *)
    int dist_group_score_data(const dist_shared_t *, const uint32_t *, float *)
    ctypedef struct dist_dpd_shared_t:
        pass
    dist_dpd_shared_t * dist_dpd_shared_create()
    void dist_dpd_shared_destroy(dist_dpd_shared_t *)
    int dist_dpd_shared_load(dist_dpd_shared_t *, float, float,
                             const uint32_t *, const float *, const int *,
                             size_t)
    int dist_dpd_shared_add_value(dist_dpd_shared_t *, uint32_t, uint32_t *)
    int dist_dpd_shared_remove_value(dist_dpd_shared_t *, uint32_t)
    int dist_dpd_shared_realize(dist_dpd_shared_t *, uint32_t *)
    size_t dist_dpd_shared_slots(const dist_dpd_shared_t *)
    size_t dist_dpd_shared_size(const dist_dpd_shared_t *)
    uint64_t dist_dpd_shared_version(const dist_dpd_shared_t *)
    int dist_dpd_shared_params(const dist_dpd_shared_t *, float *, float *,
                               float *)
    int dist_dpd_shared_view(const dist_dpd_shared_t *, dist_shared_t *)
    int dist_dpd_shared_slot(const dist_dpd_shared_t *, uint32_t, uint32_t *)
    int dist_dpd_shared_dump(const dist_dpd_shared_t *, uint32_t *, float *,
                             int *)
    int dist_sample_gamma(uint32_t *, float, float, float *)
    int dist_sample_beta_safe(uint32_t *, float, float, float, float *)
    size_t dist_scorer_words(const dist_shared_t *)
    int dist_scorer_init(const dist_shared_t *, const uint32_t *, float *)
    int dist_scorer_eval(const dist_shared_t *, const float *, uint32_t,
                         float *)
    int dist_group_protobuf_dump(const dist_shared_t *, const uint32_t *,
                                 const uint32_t *, uint8_t *, size_t, size_t *)
    int dist_group_protobuf_load(const dist_shared_t *, const uint32_t *,
                                 const uint8_t *, size_t, uint32_t *)
    int dist_shared_protobuf_dump(const dist_shared_t *, uint8_t *, size_t,
                                  size_t *)
    int dist_shared_protobuf_load(int, const uint8_t *, size_t,
                                  dist_shared_t *)

    ctypedef struct dist_id_tracker_t:
        pass
    dist_id_tracker_t * dist_id_tracker_create()
    void dist_id_tracker_destroy(dist_id_tracker_t *)
    int dist_id_tracker_init(dist_id_tracker_t *, size_t)
    int dist_id_tracker_add_group(dist_id_tracker_t *)
    int dist_id_tracker_remove_group(dist_id_tracker_t *, uint32_t)
    int dist_id_tracker_packed_to_global(const dist_id_tracker_t *, uint32_t,
                                         uint32_t *)
    int dist_id_tracker_global_to_packed(const dist_id_tracker_t *, uint32_t,
                                         uint32_t *)
    size_t dist_id_tracker_packed_size(const dist_id_tracker_t *)
    size_t dist_id_tracker_global_size(const dist_id_tracker_t *)

    ctypedef struct dist_gibbs_t:
        pass
    dist_gibbs_t * dist_gibbs_create(float, float, int, const dist_shared_t *)
    dist_gibbs_t * dist_gibbs_create_low_entropy(int, int,
                                                 const dist_shared_t *)
    ctypedef struct dist_comm_t:
        pass
    int dist_comm_available()
    int dist_comm_unique_id(uint8_t *)
    int dist_comm_unique_id_host(uint8_t *)
    dist_comm_t * dist_comm_create(const uint8_t *, int, int)
    int dist_comm_size(const dist_comm_t *, int *, int *)
    int dist_comm_all_reduce_dev(dist_comm_t *, void *, size_t, int, int)
    int dist_gibbs_partition_by_value(dist_gibbs_t *, dist_comm_t *)
    int dist_gibbs_gather_cells(dist_gibbs_t *, dist_comm_t *)
    int dist_gibbs_comm_volume(dist_gibbs_t *, uint64_t *, int)
    void dist_comm_destroy(dist_comm_t *)
    int dist_gibbs_sweep_sharded(dist_gibbs_t *, dist_comm_t *, size_t, size_t,
                                 uint32_t, uint64_t)
    void dist_gibbs_destroy(dist_gibbs_t *)
    int dist_gibbs_load_rows(dist_gibbs_t *, size_t, const uint32_t * const *,
                             const uint32_t *, int, int, uint64_t)
    int dist_gibbs_load_rows_dev(dist_gibbs_t *, size_t,
                                 const uint32_t * const *, uint32_t *, int,
                                 int, uint64_t)
    int dist_gibbs_load_rows_unassigned(dist_gibbs_t *, size_t,
                                        const uint32_t * const *, int,
                                        uint64_t)
    int dist_gibbs_init_sequential(dist_gibbs_t *, size_t, size_t,
                                   uint32_t *, int) nogil
    size_t dist_gibbs_stat_words(const dist_gibbs_t *)
    int dist_gibbs_export_stats_dev(const dist_gibbs_t *, int32_t *)
    int dist_gibbs_import_stats_dev(dist_gibbs_t *, const int32_t *)
    int dist_gibbs_sweep(dist_gibbs_t *, size_t, size_t, size_t, uint32_t,
                         uint64_t)
    int dist_gibbs_sweep_sequential(dist_gibbs_t *, size_t, size_t, uint32_t *)
    int dist_gibbs_sweep_sequential_many(dist_gibbs_t * const *, size_t,
                                         size_t, size_t, uint32_t *) nogil
    int dist_gibbs_batch_sample(dist_gibbs_t *, size_t, size_t, uint32_t,
                                uint64_t)
    int dist_gibbs_batch_delta_dev(dist_gibbs_t *, int32_t *)
    int dist_gibbs_batch_apply_delta_dev(dist_gibbs_t *, const int32_t *)
    int dist_gibbs_ordered_features(const dist_gibbs_t *, int *)
    int dist_gibbs_batch_moves_dev(dist_gibbs_t *, uint32_t *, uint32_t *)
    int dist_gibbs_replay_ordered_dev(dist_gibbs_t *, const uint32_t *,
                                      const uint32_t *,
                                      const uint32_t * const *, size_t, int)
    int dist_gibbs_batch_apply_local(dist_gibbs_t *)
    int dist_gibbs_batch_finish(dist_gibbs_t *)
    int dist_gibbs_row_scores(dist_gibbs_t *, size_t, float *, size_t *)
    int dist_gibbs_score_rows_dev(dist_gibbs_t *, size_t, size_t, float *,
                                  size_t)
    size_t dist_gibbs_group_count(const dist_gibbs_t *)
    size_t dist_gibbs_row_count(const dist_gibbs_t *)
    int dist_gibbs_counts(const dist_gibbs_t *, int *)
    int dist_gibbs_assignments(const dist_gibbs_t *, uint32_t *)
    int dist_gibbs_get_group(const dist_gibbs_t *, int, size_t, uint32_t *)
    int dist_gibbs_packed_to_global(const dist_gibbs_t *, uint32_t, uint32_t *)
    int dist_gibbs_global_to_packed(const dist_gibbs_t *, uint32_t, uint32_t *)
    size_t dist_gibbs_global_size(const dist_gibbs_t *)
    int dist_gibbs_debug_counts(dist_gibbs_t *, uint64_t *, size_t)
    ctypedef struct dist_validate_report_t:
        int code
        int feature
        long long group
        long long detail
        long long expected
        long long found
        long long rows_assigned
        char what[96]
    int dist_gibbs_validate(dist_gibbs_t *, dist_validate_report_t *)
    int dist_gibbs_comm_stats(dist_gibbs_t *, double *, uint64_t *, int)
    int dist_gibbs_phase_stats(dist_gibbs_t *, double *, uint64_t *, int)
    size_t dist_gibbs_float_delta_words(const dist_gibbs_t *)
    int dist_gibbs_batch_float_delta_dev(dist_gibbs_t *, double *)
    int dist_gibbs_batch_apply_float_delta_dev(dist_gibbs_t *, const double *)
    int dist_gibbs_export_float_moments_dev(dist_gibbs_t *, double *)
    int dist_gibbs_import_float_moments_dev(dist_gibbs_t *, const double *)
    int dist_gibbs_sharded_device_normalise_ok(const dist_gibbs_t *, size_t,
                                               size_t, int *)
    int dist_gibbs_kernel_stats(dist_gibbs_t *, double *, uint64_t *,
                                uint64_t *, int)
    int dist_gibbs_set_option(dist_gibbs_t *, const char *, int)
    int dist_gibbs_path_counts(const dist_gibbs_t *, uint64_t *, uint64_t *)


KIND_DD = DIST_DD
KIND_BB = DIST_BB
KIND_GP = DIST_GP
KIND_NICH = DIST_NICH
KIND_DPD = DIST_DPD
KIND_BNB = DIST_BNB
DPD_OTHER = 0xFFFFFFFF


cdef inline int check(int rc) except -1:
    if rc != 0:
        raise RuntimeError(dist_last_error().decode("utf-8", "replace"))
    return 0


cdef inline object checked_size(size_t n):
    # the size_t getters report a failure as (size_t)-1
    if n == <size_t> -1:
        raise RuntimeError(dist_last_error().decode("utf-8", "replace"))
    return n


def abi_version():
    return dist_abi_version()


def device_count():
    cdef int n = 0
    check(dist_device_count(&n))
    return n


def set_device(int device):
    check(dist_set_device(device))


def synchronize():
    check(dist_synchronize())


def set_stream(size_t hip_stream):
    """the calling thread's HIP stream (0 = default), e.g.
    torch.cuda.Stream().cuda_stream"""
    check(dist_set_stream(<void *> hip_stream))


# ---------------------------------------------------------------------------
# Shared: the hyper-parameters of one feature

cdef class SharedParams:
    """dist_shared_t plus the storage it points at."""
    cdef dist_shared_t c
    cdef object _betas

    def __cinit__(self):
        memset(&self.c, 0, sizeof(dist_shared_t))
        self._betas = None

    @staticmethod
    def make(int kind, p=(), alphas=(), betas=None):
        cdef SharedParams s = SharedParams()
        cdef int i
        cdef cnp.ndarray[cnp.float32_t, ndim=1] b
        s.c.kind = kind
        for i in range(len(p)):
            s.c.p[i] = p[i]
        if kind == DIST_DD:
            if not 1 <= len(alphas) <= 256:
                raise RuntimeError("expected 1 <= dim <= 256")
            s.c.dim = len(alphas)
            for i in range(len(alphas)):
                s.c.alphas[i] = alphas[i]
        if kind == DIST_DPD:
            b = np.ascontiguousarray(betas, dtype=np.float32)
            s._betas = b
            s.c.dim = b.shape[0]
            s.c.betas = <const float *> b.data
        return s

    property kind:
        def __get__(self):
            return self.c.kind

    property dim:
        def __get__(self):
            return self.c.dim

    property p:
        def __get__(self):
            return [self.c.p[i] for i in range(4)]

    property alphas:
        def __get__(self):
            return [self.c.alphas[i] for i in range(self.c.dim)]

    property betas:
        def __get__(self):
            return None if self._betas is None else self._betas.copy()

    def group_words(self):
        return dist_group_words(&self.c)

    # Model::Group scalar operations on a words array
    def group_init(self):
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] g = np.zeros(
            self.group_words(), np.uint32)
        check(dist_group_init(&self.c, <uint32_t *> g.data))
        return g

    def group_add_value(self, cnp.ndarray[cnp.uint32_t, ndim=1] g,
                        uint32_t value):
        check(dist_group_add_value(&self.c, <uint32_t *> g.data, value))

    def group_remove_value(self, cnp.ndarray[cnp.uint32_t, ndim=1] g,
                           uint32_t value):
        check(dist_group_remove_value(&self.c, <uint32_t *> g.data, value))

    def group_score_value(self, cnp.ndarray[cnp.uint32_t, ndim=1] g,
                          uint32_t value):
        cdef float out = 0
        check(dist_group_score_value(&self.c, <const uint32_t *> g.data, value,
                                     &out))
        return out


    def group_score_data(self, cnp.ndarray[cnp.uint32_t, ndim=1] g):
        cdef float out = 0
        check(dist_group_score_data(&self.c, <const uint32_t *> g.data, &out))
        return out

    # Model::Scorer (dd.hpp:222-245 and the other models'): init -> state
    def scorer_init(self, cnp.ndarray[cnp.uint32_t, ndim=1] g):
        cdef cnp.ndarray[cnp.float32_t, ndim=1] state = np.zeros(
            dist_scorer_words(&self.c), np.float32)
        check(dist_scorer_init(&self.c, <const uint32_t *> g.data,
                               <float *> state.data))
        return state

    def scorer_eval(self, cnp.ndarray[cnp.float32_t, ndim=1] state,
                    uint32_t value):
        cdef float out = 0
        check(dist_scorer_eval(&self.c, <const float *> state.data, value,
                               &out))
        return out

    # protobuf wire bytes (schema.proto messages) without libprotobuf
    def group_protobuf_dump(self, cnp.ndarray[cnp.uint32_t, ndim=1] g,
                            keys=None):
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] k
        cdef const uint32_t * kp = NULL
        if keys is not None:
            k = np.ascontiguousarray(keys, np.uint32)
            kp = <const uint32_t *> k.data
        cdef size_t n = 0
        check(dist_group_protobuf_dump(&self.c, <const uint32_t *> g.data, kp,
                                       NULL, 0, &n))
        cdef cnp.ndarray[cnp.uint8_t, ndim=1] buf = np.zeros(max(n, 1),
                                                            np.uint8)
        check(dist_group_protobuf_dump(&self.c, <const uint32_t *> g.data, kp,
                                       <uint8_t *> buf.data, n, &n))
        return buf[:n].tobytes()

    def group_protobuf_load(self, bytes data, keys=None):
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] k
        cdef const uint32_t * kp = NULL
        if keys is not None:
            k = np.ascontiguousarray(keys, np.uint32)
            kp = <const uint32_t *> k.data
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] g = np.zeros(
            self.group_words(), np.uint32)
        cdef const unsigned char * d = data
        check(dist_group_protobuf_load(&self.c, kp, <const uint8_t *> d,
                                       len(data), <uint32_t *> g.data))
        return g

    def protobuf_dump(self):
        cdef size_t n = 0
        check(dist_shared_protobuf_dump(&self.c, NULL, 0, &n))
        cdef cnp.ndarray[cnp.uint8_t, ndim=1] buf = np.zeros(max(n, 1),
                                                            np.uint8)
        check(dist_shared_protobuf_dump(&self.c, <uint8_t *> buf.data, n, &n))
        return buf[:n].tobytes()

    @staticmethod
    def protobuf_load(int kind, bytes data):
        cdef SharedParams s = SharedParams()
        cdef const unsigned char * d = data
        check(dist_shared_protobuf_load(kind, <const uint8_t *> d, len(data),
                                        &s.c))
        return s


# ---------------------------------------------------------------------------
# entropy and sampling

cdef class DpdShared:
    """DirichletProcessDiscrete::Shared's stick-breaking state
    (dist_dpd_shared_t; dpd.hpp:59-124)"""
    cdef dist_dpd_shared_t * h

    def __cinit__(self):
        self.h = dist_dpd_shared_create()
        if self.h == NULL:
            raise MemoryError()

    def __dealloc__(self):
        if self.h != NULL:
            dist_dpd_shared_destroy(self.h)
            self.h = NULL

    def load(self, float gamma, float alpha, values, betas, counts):
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] v = np.ascontiguousarray(
            values, dtype=np.uint32)
        cdef cnp.ndarray[cnp.float32_t, ndim=1] b = np.ascontiguousarray(
            betas, dtype=np.float32)
        cdef cnp.ndarray[cnp.int32_t, ndim=1] c = np.ascontiguousarray(
            counts, dtype=np.int32)
        if not (v.shape[0] == b.shape[0] == c.shape[0]):
            raise RuntimeError("invalid message")
        check(dist_dpd_shared_load(self.h, gamma, alpha,
                                   <const uint32_t *> v.data,
                                   <const float *> b.data,
                                   <const int *> c.data, v.shape[0]))

    def add_value(self, uint32_t value, uint32_t rng_state):
        check(dist_dpd_shared_add_value(self.h, value, &rng_state))
        return rng_state

    def remove_value(self, uint32_t value):
        check(dist_dpd_shared_remove_value(self.h, value))

    def realize(self, uint32_t rng_state):
        check(dist_dpd_shared_realize(self.h, &rng_state))
        return rng_state

    def slot(self, uint32_t value):
        cdef uint32_t out = 0
        check(dist_dpd_shared_slot(self.h, value, &out))
        return out

    def __len__(self):
        return dist_dpd_shared_size(self.h)

    property slots:
        def __get__(self):
            return dist_dpd_shared_slots(self.h)

    property version:
        def __get__(self):
            return dist_dpd_shared_version(self.h)

    def scalars(self):
        """(gamma, alpha, beta0)"""
        cdef float g = 0, a = 0, b = 0
        check(dist_dpd_shared_params(self.h, &g, &a, &b))
        return g, a, b

    def dump(self):
        """by dense slot: values (0xFFFFFFFF = free), betas, counts"""
        cdef size_t n = dist_dpd_shared_slots(self.h)
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] v = np.zeros(n, np.uint32)
        cdef cnp.ndarray[cnp.float32_t, ndim=1] b = np.zeros(n, np.float32)
        cdef cnp.ndarray[cnp.int32_t, ndim=1] c = np.zeros(n, np.int32)
        check(dist_dpd_shared_dump(self.h, <uint32_t *> v.data,
                                   <float *> b.data, <int *> c.data))
        return v, b, c

    def params(self):
        """the SharedParams (dist_shared_t) mixtures and groups take: a
        snapshot (betas copied) of the dense view"""
        cdef dist_shared_t view
        check(dist_dpd_shared_view(self.h, &view))
        v, b, c = self.dump()
        return SharedParams.make(DIST_DPD, p=(view.p[0], view.p[1]), betas=b)


def sample_gamma(uint32_t rng_state, float alpha, float beta=1.0):
    """random.hpp:87-97 -> (draw, engine state after)"""
    cdef float out = 0
    check(dist_sample_gamma(&rng_state, alpha, beta, &out))
    return out, rng_state


def sample_beta_safe(uint32_t rng_state, float alpha, float beta,
                     float min_value):
    """random.hpp:110-119 -> (draw, engine state after)"""
    cdef float out = 0
    check(dist_sample_beta_safe(&rng_state, alpha, beta, min_value, &out))
    return out, rng_state


def rng_seed(seed):
    return dist_rng_seed(<uint64_t> seed)


def rng_next(uint32_t state):
    """-> (raw engine output == new state)"""
    cdef uint32_t s = state
    dist_rng_next(&s)
    return s


def rng_unif01(uint32_t state):
    """-> (u, new state)"""
    cdef uint32_t s = state
    cdef float u = dist_rng_unif01(&s)
    return u, s


def rng_jump(uint32_t state, steps):
    return dist_rng_jump(state, <uint64_t> steps)


def _vector(int op, x):
    cdef cnp.ndarray[cnp.float32_t, ndim=1] a
    cdef cnp.ndarray[cnp.uint32_t, ndim=1] u
    cdef cnp.ndarray[cnp.float32_t, ndim=1] out
    if op == 4:
        u = np.ascontiguousarray(x, dtype=np.uint32)
        out = np.zeros(u.shape[0], np.float32)
        check(dist_vector_log_factorial(u.shape[0], <const uint32_t *> u.data,
                                        <float *> out.data))
        return out
    a = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(a.shape[0], np.float32)
    if op == 0:
        check(dist_vector_log(a.shape[0], <const float *> a.data, <float *> out.data))
    elif op == 1:
        check(dist_vector_exp(a.shape[0], <const float *> a.data, <float *> out.data))
    elif op == 2:
        check(dist_vector_lgamma(a.shape[0], <const float *> a.data, <float *> out.data))
    else:
        check(dist_vector_lgamma_nu(a.shape[0], <const float *> a.data, <float *> out.data))
    return out


def vector_log(x):
    return _vector(0, x)


def vector_exp(x):
    return _vector(1, x)


def vector_lgamma(x):
    return _vector(2, x)


def vector_lgamma_nu(x):
    return _vector(3, x)


def vector_log_factorial(x):
    return _vector(4, x)


def sample_from_scores_overwrite(uint32_t state,
                                 cnp.ndarray[cnp.float32_t, ndim=1] scores):
    """-> (sample, new state); scores become likelihoods in place"""
    cdef uint32_t s = state
    cdef size_t sample = 0
    check(dist_sample_from_scores_overwrite(&s, scores.shape[0],
                                            <float *> scores.data, &sample))
    return sample, s


def scores_to_likelihoods(cnp.ndarray[cnp.float32_t, ndim=1] scores):
    cdef float total = 0
    check(dist_scores_to_likelihoods(scores.shape[0], <float *> scores.data,
                                     &total))
    return total


def sample_from_likelihoods(uint32_t state,
                            cnp.ndarray[cnp.float32_t, ndim=1] likelihoods,
                            float total):
    cdef uint32_t s = state
    cdef size_t sample = 0
    check(dist_sample_from_likelihoods(&s, likelihoods.shape[0],
                                       <const float *> likelihoods.data, total,
                                       &sample))
    return sample, s


def log_sum_exp(scores):
    cdef cnp.ndarray[cnp.float32_t, ndim=1] a = np.ascontiguousarray(
        scores, dtype=np.float32)
    cdef float out = 0
    check(dist_log_sum_exp(a.shape[0], <const float *> a.data, &out))
    return out


def py_score_add_value(float alpha, float d, int group_size,
                       int nonempty_group_count, int sample_size,
                       int empty_group_count=1):
    cdef float out = 0
    check(dist_py_score_add_value(alpha, d, group_size, nonempty_group_count,
                                  sample_size, empty_group_count, &out))
    return out


def py_score_remove_value(float alpha, float d, int group_size,
                          int nonempty_group_count, int sample_size,
                          int empty_group_count=1):
    cdef float out = 0
    check(dist_py_score_remove_value(alpha, d, group_size,
                                     nonempty_group_count, sample_size,
                                     empty_group_count, &out))
    return out


def py_sample_assignments(float alpha, float d, int size, uint32_t state):
    """-> (assignments, new rng state)"""
    cdef cnp.ndarray[cnp.int32_t, ndim=1] out = np.zeros(max(size, 1),
                                                         np.int32)
    cdef uint32_t s = state
    check(dist_py_sample_assignments(alpha, d, size, &s, <int *> out.data))
    return out[:size], s


def py_score_counts(float alpha, float d, counts):
    cdef cnp.ndarray[cnp.int32_t, ndim=1] c = np.ascontiguousarray(
        counts, dtype=np.int32)
    cdef float out = 0
    check(dist_py_score_counts(alpha, d, <const int *> c.data, c.shape[0],
                               &out))
    return out


# ---------------------------------------------------------------------------
# PitmanYor::Mixture

cdef class PyMixture:
    cdef dist_py_mixture_t * ptr

    def __cinit__(self):
        self.ptr = dist_py_mixture_create()
        if self.ptr == NULL:
            raise RuntimeError(dist_last_error().decode())

    def __dealloc__(self):
        if self.ptr != NULL:
            dist_py_mixture_destroy(self.ptr)

    def __len__(self):
        return dist_py_mixture_size(self.ptr)

    def init(self, float alpha, float d, counts):
        cdef cnp.ndarray[cnp.int32_t, ndim=1] c = np.ascontiguousarray(
            counts, dtype=np.int32)
        check(dist_py_mixture_init(self.ptr, alpha, d, <const int *> c.data,
                                   c.shape[0]))

    def add_value(self, float alpha, float d, size_t groupid):
        cdef int added = 0
        check(dist_py_mixture_add_value(self.ptr, alpha, d, groupid, &added))
        return bool(added)

    def remove_value(self, float alpha, float d, size_t groupid):
        cdef int removed = 0
        check(dist_py_mixture_remove_value(self.ptr, alpha, d, groupid,
                                           &removed))
        return bool(removed)

    def score_value(self, float alpha, float d,
                    cnp.ndarray[cnp.float32_t, ndim=1] scores):
        check(dist_py_mixture_score_value(self.ptr, alpha, d,
                                          <float *> scores.data,
                                          scores.shape[0]))

    def score_data(self, float alpha, float d):
        cdef float out = 0
        check(dist_py_mixture_score_data(self.ptr, alpha, d, &out))
        return out

    def sample_size(self):
        return dist_py_mixture_sample_size(self.ptr)

    def counts(self):
        cdef cnp.ndarray[cnp.int32_t, ndim=1] out = np.zeros(len(self),
                                                            np.int32)
        if len(self):
            check(dist_py_mixture_counts(self.ptr, <int *> out.data))
        return out

    def empty_groupids(self):
        cdef size_t n = dist_py_mixture_empty_groupids(self.ptr, NULL, 0)
        cdef size_t * buf = <size_t *> malloc(sizeof(size_t) * (n + 1))
        dist_py_mixture_empty_groupids(self.ptr, buf, n)
        out = [buf[i] for i in range(n)]
        free(buf)
        return out


# ---------------------------------------------------------------------------
# Clustering<int>::LowEntropy

def le_score_add_value(int dataset_size, int group_size, int nonempty,
                       int sample_size, int empty=1):
    cdef float out = 0
    check(dist_le_score_add_value(dataset_size, group_size, nonempty,
                                  sample_size, empty, &out))
    return out


def le_score_remove_value(int dataset_size, int group_size, int nonempty,
                          int sample_size, int empty=1):
    cdef float out = 0
    check(dist_le_score_remove_value(dataset_size, group_size, nonempty,
                                     sample_size, empty, &out))
    return out


def le_log_partition_function(int sample_size):
    cdef float out = 0
    check(dist_le_log_partition_function(sample_size, &out))
    return out


def le_sample_assignments(int dataset_size, int size, uint32_t rng_state):
    """-> (int32 assignments, new rng state)"""
    cdef cnp.ndarray[cnp.int32_t, ndim=1] out = np.zeros(max(size, 1), np.int32)
    cdef uint32_t s = rng_state
    check(dist_le_sample_assignments(dataset_size, size, &s, <int *> out.data))
    return out[:size], s


def le_score_counts(int dataset_size, counts):
    cdef cnp.ndarray[cnp.int32_t, ndim=1] c = np.ascontiguousarray(
        counts, dtype=np.int32)
    cdef float out = 0
    check(dist_le_score_counts(dataset_size, <const int *> c.data, c.shape[0],
                               &out))
    return out


cdef class LeMixture:
    cdef dist_le_mixture_t * ptr

    def __cinit__(self):
        self.ptr = dist_le_mixture_create()
        if self.ptr == NULL:
            raise RuntimeError(dist_last_error().decode())

    def __dealloc__(self):
        if self.ptr != NULL:
            dist_le_mixture_destroy(self.ptr)

    def __len__(self):
        return dist_le_mixture_size(self.ptr)

    def init(self, counts):
        cdef cnp.ndarray[cnp.int32_t, ndim=1] c = np.ascontiguousarray(
            counts, dtype=np.int32)
        check(dist_le_mixture_init(self.ptr, <const int *> c.data, c.shape[0]))

    def add_value(self, size_t groupid):
        cdef int added = 0
        check(dist_le_mixture_add_value(self.ptr, groupid, &added))
        return bool(added)

    def remove_value(self, size_t groupid):
        cdef int removed = 0
        check(dist_le_mixture_remove_value(self.ptr, groupid, &removed))
        return bool(removed)

    def score_value(self, int dataset_size,
                    cnp.ndarray[cnp.float32_t, ndim=1] scores):
        check(dist_le_mixture_score_value(self.ptr, dataset_size,
                                          <float *> scores.data,
                                          scores.shape[0]))

    def score_data(self, int dataset_size):
        cdef float out = 0
        check(dist_le_mixture_score_data(self.ptr, dataset_size, &out))
        return out

    def sample_size(self):
        return dist_le_mixture_sample_size(self.ptr)

    def counts(self):
        cdef cnp.ndarray[cnp.int32_t, ndim=1] out = np.zeros(len(self),
                                                            np.int32)
        if len(self):
            check(dist_le_mixture_counts(self.ptr, <int *> out.data))
        return out

    def empty_groupids(self):
        return [i for i, c in enumerate(self.counts()) if c == 0]


# ---------------------------------------------------------------------------
# Model::Mixture

cdef class SlaveMixture:
    cdef dist_mixture_t * ptr
    cdef SharedParams shared

    def __cinit__(self, SharedParams shared):
        self.shared = shared
        self.ptr = dist_mixture_create(&shared.c)
        if self.ptr == NULL:
            raise RuntimeError(dist_last_error().decode())

    def __dealloc__(self):
        if self.ptr != NULL:
            dist_mixture_destroy(self.ptr)

    def __len__(self):
        return dist_mixture_size(self.ptr)

    def clear(self):
        check(dist_mixture_clear(self.ptr))

    def append(self, cnp.ndarray[cnp.uint32_t, ndim=1] group):
        if group.shape[0] != <Py_ssize_t> self.shared.group_words():
            raise RuntimeError("group has the wrong size")
        check(dist_mixture_append(self.ptr, <const uint32_t *> group.data))

    def get_group(self, size_t groupid):
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] g = np.zeros(
            self.shared.group_words(), np.uint32)
        check(dist_mixture_get_group(self.ptr, groupid, <uint32_t *> g.data))
        return g

    def init(self):
        check(dist_mixture_init(self.ptr))

    def add_group(self):
        check(dist_mixture_add_group(self.ptr))

    def remove_group(self, size_t groupid):
        check(dist_mixture_remove_group(self.ptr, groupid))

    def add_value(self, size_t groupid, uint32_t value):
        check(dist_mixture_add_value(self.ptr, groupid, value))

    def remove_value(self, size_t groupid, uint32_t value):
        check(dist_mixture_remove_value(self.ptr, groupid, value))

    def score_value_group(self, size_t groupid, uint32_t value):
        cdef float out = 0
        check(dist_mixture_score_value_group(self.ptr, groupid, value, &out))
        return out

    def score_value(self, uint32_t value,
                    cnp.ndarray[cnp.float32_t, ndim=1] scores_accum):
        check(dist_mixture_score_value(self.ptr, value,
                                       <float *> scores_accum.data,
                                       scores_accum.shape[0]))

    def score_data(self):
        cdef float out = 0
        check(dist_mixture_score_data(self.ptr, &out))
        return out

    def score_values(self, values,
                     cnp.ndarray[cnp.float32_t, ndim=2, mode="c"] scores_accum):
        """score_value for len(values) values in one launch:
        scores_accum[r, k] accumulates (dist_mixture_score_values)"""
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] v = np.ascontiguousarray(
            values, dtype=np.uint32)
        if scores_accum.shape[0] != v.shape[0]:
            raise RuntimeError("scores_accum has one row per value")
        check(dist_mixture_score_values(self.ptr, <const uint32_t *> v.data,
                                        v.shape[0],
                                        <float *> scores_accum.data,
                                        scores_accum.shape[1]))

    def validate(self):
        """Mixture::validate (mixture.hpp:440-444): raises RuntimeError"""
        check(dist_mixture_validate(self.ptr))

    def score_data_grid(self, shareds):
        """shareds: SharedParams candidates -> float32 scores, one each"""
        cdef size_t n = len(shareds)
        cdef cnp.ndarray[cnp.float32_t, ndim=1] out = np.zeros(n, np.float32)
        if n == 0:
            return out
        cdef dist_shared_t * arr = <dist_shared_t *> malloc(
            n * sizeof(dist_shared_t))
        cdef SharedParams s
        for i in range(n):
            s = shareds[i]
            arr[i] = s.c
        cdef int rc = dist_mixture_score_data_grid(self.ptr, arr, n,
                                                   <float *> out.data)
        free(arr)
        check(rc)
        return out


# ---------------------------------------------------------------------------
# MixtureIdTracker

cdef class IdTracker:
    cdef dist_id_tracker_t * ptr

    def __cinit__(self):
        self.ptr = dist_id_tracker_create()

    def __dealloc__(self):
        if self.ptr != NULL:
            dist_id_tracker_destroy(self.ptr)

    def init(self, size_t group_count=0):
        check(dist_id_tracker_init(self.ptr, group_count))

    def add_group(self):
        check(dist_id_tracker_add_group(self.ptr))

    def remove_group(self, uint32_t packed):
        check(dist_id_tracker_remove_group(self.ptr, packed))

    def packed_to_global(self, uint32_t packed):
        cdef uint32_t out = 0
        check(dist_id_tracker_packed_to_global(self.ptr, packed, &out))
        return out

    def global_to_packed(self, uint32_t global_):
        cdef uint32_t out = 0
        check(dist_id_tracker_global_to_packed(self.ptr, global_, &out))
        return out

    def packed_size(self):
        return dist_id_tracker_packed_size(self.ptr)

    def global_size(self):
        return dist_id_tracker_global_size(self.ptr)


# ---------------------------------------------------------------------------
# the batched row engine

def value_words(int kind, values):
    """Values of one feature as the 32-bit words the ABI takes."""
    if kind == DIST_NICH:
        return np.ascontiguousarray(values, dtype=np.float32).view(np.uint32)
    return np.ascontiguousarray(values).astype(np.uint32)


def comm_available():
    """True if RCCL can be bound at run time"""
    return bool(dist_comm_available())


def comm_unique_id():
    """128 bytes for rank 0 to hand to the other ranks"""
    cdef cnp.ndarray[cnp.uint8_t, ndim=1] out = np.zeros(128, np.uint8)
    check(dist_comm_unique_id(<uint8_t *> out.data))
    return out


def comm_unique_id_host():
    """128 bytes naming a host (shared-memory) communicator: ranks that share
    one GPU, where RCCL cannot be used"""
    cdef cnp.ndarray[cnp.uint8_t, ndim=1] out = np.zeros(128, np.uint8)
    check(dist_comm_unique_id_host(<uint8_t *> out.data))
    return out


cdef class Comm:
    """the library's own communicator (collective constructor): RCCL, or the
    host transport when the id came from comm_unique_id_host()"""
    cdef dist_comm_t * ptr

    def __cinit__(self, unique_id, int rank, int world):
        cdef cnp.ndarray[cnp.uint8_t, ndim=1] uid = np.ascontiguousarray(
            unique_id, dtype=np.uint8)
        if uid.shape[0] != 128:
            raise ValueError("the unique id is 128 bytes")
        cdef const uint8_t * p = <const uint8_t *> uid.data
        with nogil:
            self.ptr = dist_comm_create(p, rank, world)
        if self.ptr == NULL:
            raise RuntimeError(dist_last_error().decode())

    def __dealloc__(self):
        if self.ptr != NULL:
            dist_comm_destroy(self.ptr)

    def size(self):
        """-> (rank, world)"""
        cdef int r = 0, w = 0
        check(dist_comm_size(self.ptr, &r, &w))
        return r, w

    def all_reduce_dev(self, size_t ptr, size_t count, dtype="int32",
                       op="sum"):
        """in place, on device memory; waited for"""
        cdef int t = {"int32": 0, "float64": 1}[dtype]
        cdef int o = {"sum": 0, "min": 1}[op]
        cdef int rc
        with nogil:
            rc = dist_comm_all_reduce_dev(self.ptr, <void *> ptr, count, t, o)
        check(rc)


cdef class GibbsEngine:
    cdef dist_gibbs_t * ptr
    cdef list shareds
    cdef object _keep   # device tensors / arrays the engine points into

    def __cinit__(self, float alpha, float d, shareds, dataset_size=None):
        """dataset_size: the LowEntropy clustering model instead of
        PitmanYor(alpha, d)"""
        cdef int n = len(shareds)
        cdef dist_shared_t * arr = <dist_shared_t *> malloc(
            sizeof(dist_shared_t) * (n + 1))
        cdef SharedParams s
        cdef int i
        for i in range(n):
            s = shareds[i]
            memcpy(&arr[i], &s.c, sizeof(dist_shared_t))
        self.shareds = list(shareds)
        if dataset_size is None:
            self.ptr = dist_gibbs_create(alpha, d, n, arr)
        else:
            self.ptr = dist_gibbs_create_low_entropy(int(dataset_size), n, arr)
        free(arr)
        if self.ptr == NULL:
            raise RuntimeError(dist_last_error().decode())

    def __dealloc__(self):
        if self.ptr != NULL:
            dist_gibbs_destroy(self.ptr)

    def load_rows(self, values, assign_packed, int nonempty_groups,
                  int empty_groups=1, row_offset=0):
        """values: one host array per feature; assign_packed: host uint32."""
        cdef int n = len(self.shareds)
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] a = np.ascontiguousarray(
            assign_packed, dtype=np.uint32)
        cdef const uint32_t ** ptrs = <const uint32_t **> malloc(
            sizeof(void *) * (n + 1))
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] w
        cdef SharedParams s
        words = []
        cdef int i
        for i in range(n):
            s = self.shareds[i]
            w = value_words(s.c.kind, values[i])
            if w.shape[0] != a.shape[0]:
                free(ptrs)
                raise RuntimeError("feature columns differ in length")
            words.append(w)
            ptrs[i] = <const uint32_t *> w.data
        cdef int rc = dist_gibbs_load_rows(self.ptr, a.shape[0], ptrs,
                                           <const uint32_t *> a.data,
                                           nonempty_groups, empty_groups,
                                           <uint64_t> row_offset)
        free(ptrs)
        check(rc)

    def load_rows_unassigned(self, values, int empty_groups=1, row_offset=0):
        """rows without a group yet; init_sequential assigns them"""
        cdef int n = len(self.shareds)
        cdef const uint32_t ** ptrs = <const uint32_t **> malloc(
            sizeof(void *) * (n + 1))
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] w
        cdef SharedParams s
        words = []
        cdef int i
        cdef size_t rows = 0
        for i in range(n):
            s = self.shareds[i]
            w = value_words(s.c.kind, values[i])
            if i and w.shape[0] != rows:
                free(ptrs)
                raise RuntimeError("feature columns differ in length")
            rows = w.shape[0]
            words.append(w)
            ptrs[i] = <const uint32_t *> w.data
        cdef int rc = dist_gibbs_load_rows_unassigned(
            self.ptr, rows, ptrs, empty_groups, <uint64_t> row_offset)
        free(ptrs)
        check(rc)

    def init_sequential(self, size_t row_begin, size_t row_end,
                        uint32_t rng_state, prior_only=False):
        """-> new rng state"""
        cdef uint32_t s = rng_state
        cdef int rc
        cdef int prior = 1 if prior_only else 0
        with nogil:
            rc = dist_gibbs_init_sequential(self.ptr, row_begin, row_end, &s,
                                            prior)
        check(rc)
        return s

    def load_rows_dev(self, value_ptrs, size_t assign_ptr, size_t n_rows,
                      int nonempty_groups, int empty_groups=1, row_offset=0,
                      keep=None):
        """Device pointers (e.g. torch tensor data_ptr()); `keep` holds the
        owners alive for the lifetime of the engine."""
        cdef int n = len(self.shareds)
        cdef const uint32_t ** ptrs = <const uint32_t **> malloc(
            sizeof(void *) * (n + 1))
        cdef int i
        cdef size_t addr
        for i in range(n):
            addr = value_ptrs[i]
            ptrs[i] = <const uint32_t *> addr
        self._keep = keep
        cdef int rc = dist_gibbs_load_rows_dev(
            self.ptr, n_rows, ptrs, <uint32_t *> assign_ptr, nonempty_groups,
            empty_groups, <uint64_t> row_offset)
        free(ptrs)
        check(rc)

    def stat_words(self):
        return checked_size(dist_gibbs_stat_words(self.ptr))

    def export_stats_dev(self, size_t ptr):
        check(dist_gibbs_export_stats_dev(self.ptr, <int32_t *> ptr))

    def import_stats_dev(self, size_t ptr):
        check(dist_gibbs_import_stats_dev(self.ptr, <const int32_t *> ptr))

    def sweep(self, size_t row_begin, size_t row_end, size_t batch_rows,
              uint32_t seed_state, draw_base=0):
        cdef uint64_t db = <uint64_t> draw_base
        cdef int rc
        with nogil:
            rc = dist_gibbs_sweep(self.ptr, row_begin, row_end, batch_rows,
                                  seed_state, db)
        check(rc)

    def sweep_sequential(self, size_t row_begin, size_t row_end,
                         uint32_t rng_state):
        """-> new rng state"""
        cdef uint32_t s = rng_state
        cdef int rc
        with nogil:
            rc = dist_gibbs_sweep_sequential(self.ptr, row_begin, row_end, &s)
        check(rc)
        return s

    def sweep_sharded(self, Comm comm, size_t n_batches, size_t batch_rows,
                      uint32_t seed_state, draw_base=0):
        cdef uint64_t db = <uint64_t> draw_base
        cdef int rc
        with nogil:
            rc = dist_gibbs_sweep_sharded(self.ptr, comm.ptr, n_batches,
                                          batch_rows, seed_state, db)
        check(rc)

    def partition_by_value(self, Comm comm):
        """collective: no value has rows on two ranks -> sweep_sharded
        exchanges 3 words per group instead of the cells"""
        cdef int rc
        with nogil:
            rc = dist_gibbs_partition_by_value(self.ptr, comm.ptr)
        check(rc)

    def gather_cells(self, Comm comm):
        """collective: make a value-partitioned replica whole again"""
        cdef int rc
        with nogil:
            rc = dist_gibbs_gather_cells(self.ptr, comm.ptr)
        check(rc)

    def comm_volume(self, reset=False):
        """-> dict: all-reduces of sweep_sharded, int32 words sent in all,
        the largest, the last"""
        cdef uint64_t v[4]
        check(dist_gibbs_comm_volume(self.ptr, v, 1 if reset else 0))
        return {"collectives": int(v[0]), "words": int(v[1]),
                "words_max": int(v[2]), "words_last": int(v[3])}

    def batch_sample(self, size_t row_begin, size_t row_end,
                     uint32_t seed_state, draw_base=0):
        check(dist_gibbs_batch_sample(self.ptr, row_begin, row_end, seed_state,
                                      <uint64_t> draw_base))

    def batch_delta_dev(self, size_t ptr):
        check(dist_gibbs_batch_delta_dev(self.ptr, <int32_t *> ptr))

    def batch_apply_delta_dev(self, size_t ptr):
        check(dist_gibbs_batch_apply_delta_dev(self.ptr, <const int32_t *> ptr))

    def batch_apply_local(self):
        check(dist_gibbs_batch_apply_local(self.ptr))

    def ordered_features(self):
        """number of features with order-dependent statistics (NICH, GP)"""
        cdef int n = 0
        check(dist_gibbs_ordered_features(self.ptr, &n))
        return n

    def feature_is_ordered(self, int feature):
        cdef SharedParams s = self.shareds[feature]
        return s.c.kind == DIST_GP or s.c.kind == DIST_NICH

    def batch_moves_dev(self, size_t old_ptr, size_t new_ptr):
        check(dist_gibbs_batch_moves_dev(self.ptr, <uint32_t *> old_ptr,
                                         <uint32_t *> new_ptr))

    def replay_ordered_dev(self, size_t old_ptr, size_t new_ptr, value_ptrs,
                           size_t n_rows, bint reset):
        """value_ptrs: per feature a device address or 0"""
        cdef int n = len(value_ptrs)
        cdef const uint32_t ** ptrs = <const uint32_t **> malloc(
            max(n, 1) * sizeof(void *))
        cdef size_t addr
        for i in range(n):
            addr = value_ptrs[i]
            ptrs[i] = <const uint32_t *> addr
        cdef int rc = dist_gibbs_replay_ordered_dev(
            self.ptr, <const uint32_t *> old_ptr, <const uint32_t *> new_ptr,
            ptrs, n_rows, reset)
        free(ptrs)
        check(rc)

    def batch_finish(self):
        check(dist_gibbs_batch_finish(self.ptr))

    def row_scores(self, size_t row):
        cdef cnp.ndarray[cnp.float32_t, ndim=1] out = np.zeros(
            self.group_count() + 1, np.float32)
        cdef size_t n = 0
        check(dist_gibbs_row_scores(self.ptr, row, <float *> out.data, &n))
        return out[:n].copy()

    def score_rows_dev(self, size_t row_begin, size_t row_end, size_t ptr,
                       size_t ld):
        check(dist_gibbs_score_rows_dev(self.ptr, row_begin, row_end,
                                        <float *> ptr, ld))

    def group_count(self):
        return checked_size(dist_gibbs_group_count(self.ptr))

    def row_count(self):
        return dist_gibbs_row_count(self.ptr)

    def counts(self):
        cdef cnp.ndarray[cnp.int32_t, ndim=1] out = np.zeros(
            self.group_count(), np.int32)
        check(dist_gibbs_counts(self.ptr, <int *> out.data))
        return out

    def assignments(self):
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] out = np.zeros(
            max(1, self.row_count()), np.uint32)
        check(dist_gibbs_assignments(self.ptr, <uint32_t *> out.data))
        return out[:self.row_count()]

    def get_group(self, int feature, size_t groupid):
        cdef SharedParams s = self.shareds[feature]
        cdef cnp.ndarray[cnp.uint32_t, ndim=1] g = np.zeros(s.group_words(),
                                                            np.uint32)
        check(dist_gibbs_get_group(self.ptr, feature, groupid,
                                   <uint32_t *> g.data))
        return g

    def packed_to_global(self, uint32_t packed):
        cdef uint32_t out = 0
        check(dist_gibbs_packed_to_global(self.ptr, packed, &out))
        return out

    def global_to_packed(self, uint32_t global_):
        cdef uint32_t out = 0
        check(dist_gibbs_global_to_packed(self.ptr, global_, &out))
        return out

    def global_size(self):
        return checked_size(dist_gibbs_global_size(self.ptr))

    def sharded_device_normalise_ok(self, size_t n_batches, size_t batch_rows):
        cdef int ok = 0
        check(dist_gibbs_sharded_device_normalise_ok(self.ptr, n_batches,
                                                     batch_rows, &ok))
        return bool(ok)

    def validate(self, raise_on_failure=True):
        """dist_gibbs_validate: the statistics against a recount from the rows
        (mixture.hpp:152-163,440-444).  Returns the report as a dict; raises
        RuntimeError on an inconsistency unless raise_on_failure is False."""
        cdef dist_validate_report_t rep
        cdef int rc = dist_gibbs_validate(self.ptr, &rep)
        if rc == 1 or (rc and raise_on_failure):
            check(rc)
        return {"code": rep.code, "feature": rep.feature, "group": rep.group,
                "detail": rep.detail, "expected": rep.expected,
                "found": rep.found, "rows_assigned": rep.rows_assigned,
                "what": (<bytes> rep.what).decode()}

    def debug_counts(self):
        """dict of the engine's path diagnostics (dist_gibbs_debug_counts)"""
        cdef uint64_t out[16]
        check(dist_gibbs_debug_counts(self.ptr, out, 16))
        return {"value_sorted_batches": out[0], "other_batches": out[1],
                "band_launches": out[2], "running_sum_launches": out[3],
                "band_values_last": out[4], "handed_over_last": out[5],
                "stream_batches": out[6], "device_normalised": out[7],
                "narrow_batches": out[8], "scratch_batches": out[9],
                "fold_batches": out[10], "scan_batches": out[11],
                "merged_batches": out[12], "fused_batches": out[13],
                "resumed_runs": out[14], "chain_launches": out[15]}

    def chain_launches(self):
        """launches of the exact-chain kernel this engine issued"""
        return self.debug_counts()["chain_launches"]

    def set_option(self, name, int value):
        check(dist_gibbs_set_option(self.ptr, name.encode(), value))

    def path_counts(self):
        """-> (batches served by the value-sorted kernel, by the generic one)"""
        cdef uint64_t a = 0, b = 0
        check(dist_gibbs_path_counts(self.ptr, &a, &b))
        return a, b

    def float_delta_words(self):
        """doubles per image of the merged float statistics (0: the ordered
        replay is in use)"""
        return checked_size(dist_gibbs_float_delta_words(self.ptr))

    def batch_float_delta_dev(self, size_t ptr):
        check(dist_gibbs_batch_float_delta_dev(self.ptr, <double *> ptr))

    def batch_apply_float_delta_dev(self, size_t ptr):
        check(dist_gibbs_batch_apply_float_delta_dev(self.ptr,
                                                     <const double *> ptr))

    def export_float_moments_dev(self, size_t ptr):
        check(dist_gibbs_export_float_moments_dev(self.ptr, <double *> ptr))

    def import_float_moments_dev(self, size_t ptr):
        check(dist_gibbs_import_float_moments_dev(self.ptr,
                                                  <const double *> ptr))

    def phase_stats(self, reset=False):
        """-> ([ms of tables, score+sample, handed-over rows, statistics,
        group set + caches], sub-sweeps timed) with option phase_timing"""
        cdef double ms[5]
        cdef uint64_t n = 0
        check(dist_gibbs_phase_stats(self.ptr, ms, &n, 1 if reset else 0))
        return [ms[i] for i in range(5)], n

    def comm_stats(self, reset=False):
        """-> (ms, count) of the timed all-reduces of sweep_sharded"""
        cdef double ms = 0
        cdef uint64_t launches = 0
        check(dist_gibbs_comm_stats(self.ptr, &ms, &launches,
                                    1 if reset else 0))
        return ms, launches

    def kernel_stats(self, reset=False):
        """-> (ms, launches, rows) of the score+sample kernel (HIP events)"""
        cdef double ms = 0
        cdef uint64_t launches = 0, rows = 0
        check(dist_gibbs_kernel_stats(self.ptr, &ms, &launches, &rows,
                                      1 if reset else 0))
        return ms, launches, rows


def sweep_sequential_many(engines, size_t row_begin, size_t row_end,
                          rng_states):
    """M independent exact chains in one launch (dist_gibbs_sweep_sequential_
    many): engines = GibbsEngine objects, rng_states = their engine states.
    -> the new states (numpy uint32)."""
    cdef size_t m = len(engines)
    cdef cnp.ndarray[cnp.uint32_t, ndim=1] st = np.ascontiguousarray(
        rng_states, dtype=np.uint32).copy()
    if <size_t> st.shape[0] != m:
        raise ValueError("one rng state per engine")
    cdef dist_gibbs_t ** ptrs = <dist_gibbs_t **> malloc(
        (m if m else 1) * sizeof(dist_gibbs_t *))
    cdef size_t i
    cdef int rc
    cdef GibbsEngine e
    try:
        for i in range(m):
            e = engines[i]
            ptrs[i] = e.ptr
        with nogil:
            rc = dist_gibbs_sweep_sequential_many(
                <dist_gibbs_t * const *> ptrs, m, row_begin, row_end,
                <uint32_t *> st.data)
    finally:
        free(ptrs)
    check(rc)
    return st
