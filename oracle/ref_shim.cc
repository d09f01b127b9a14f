// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// extern "C" probe points over the subset of the reference that compiles
// here from its own sources: src/special.cc, src/vector_math.cc,
// src/common.cc and the Eigen-free headers special.hpp, vector_math.hpp,
// vendor/fmath.hpp, mixture.hpp (MixtureDriver / MixtureIdTracker).
// Everything that includes <distributions/random.hpp> (random.cc,
// clustering.*, models/*) needs Eigen, which this image lacks, so it is
// unbuildable here (see DESIGN.md "Oracle pinning").
//
// Built by oracle/Makefile into oracle/_ref/libref.so with the reference's
// own release flags.  This file is our code; it contains no reference source.
#include <cstdint>
#include <cstring>
#include <vector>
// FastLog keeps its table private; the table is data we must pin (it depends
// on the libm/libmvec the reference was built against), so open it up here.
// (standard headers first so that only the reference header sees the macro)
#include <cmath>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <cxxabi.h>
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cstdlib>
#include <immintrin.h>
#include <distributions/common.hpp>
#include <distributions/vendor/fmath.hpp>
#define private public
#include <distributions/special.hpp>
#undef private
#include <distributions/vector_math.hpp>
#include <distributions/mixture.hpp>

namespace {
// MixtureDriver only needs a Model for its (unused here) scoring fallbacks.
struct NullModel {
    float score_add_value(int, int, int, int) const { return 0.f; }
    float score_counts(const std::vector<int> &) const { return 0.f; }
};
typedef distributions::MixtureDriver<NullModel, int> Driver;
}  // namespace

extern "C" {

// ---- special.hpp:57-89,114-171,208-214,239-273 -----------------------------
void ref_fast_log(size_t n, const float * in, float * out) {
    for (size_t i = 0; i < n; ++i) out[i] = distributions::fast_log(in[i]);
}
void ref_fast_exp(size_t n, const float * in, float * out) {
    for (size_t i = 0; i < n; ++i) out[i] = distributions::fast_exp(in[i]);
}
void ref_fast_lgamma(size_t n, const float * in, float * out) {
    for (size_t i = 0; i < n; ++i) out[i] = distributions::fast_lgamma(in[i]);
}
void ref_fast_lgamma_nu(size_t n, const float * in, float * out) {
    for (size_t i = 0; i < n; ++i)
        out[i] = distributions::fast_lgamma_nu(in[i]);
}
void ref_fast_log_factorial(size_t n, const uint32_t * in, float * out) {
    for (size_t i = 0; i < n; ++i)
        out[i] = distributions::fast_log_factorial(in[i]);
}
// coefficient tables (special.cc:144-269), exported as data
void ref_lgamma_coeff5(float * out192) {
    memcpy(out192, distributions::detail::lgamma_approx_coeff5, 192 * 4);
}
void ref_lgamma_nu_coeff3(float * out80) {
    memcpy(out80, distributions::detail::lgamma_nu_func_approx_coeff3, 80 * 4);
}
void ref_log_factorial_table(float * out64) {
    memcpy(out64, distributions::detail::log_factorial_table, 64 * 4);
}
// special.cc:35-44 FastLog(14) table as built (16384 floats)
void ref_log_table(float * out16384) {
    memcpy(out16384, distributions::detail::GLOBAL_FAST_LOG_14.table_.data(),
           16384 * 4);
}
// fmath.hpp:139-175 exp table and constants
void ref_exp_table(uint32_t * out1024, float * a, float * b) {
    const fmath::local::ExpVar<> & v = fmath::local::C<>::expVar;
    memcpy(out1024, v.tbl, 1024 * 4);
    *a = v.a[0];
    *b = v.b[0];
}

// ---- vector_math.cc --------------------------------------------------------
void ref_vector_add_subtract(size_t n, float * io, const float * a,
                             const float * b) {
    distributions::vector_add_subtract(n, io, a, b);
}
void ref_vector_add_subtract_scalar(size_t n, float * io, float a,
                                    const float * b) {
    distributions::vector_add_subtract(n, io, a, b);
}
void ref_vector_add(size_t n, float * io, const float * a) {
    distributions::vector_add(n, io, a);
}
float ref_vector_max(size_t n, const float * in) {
    return distributions::vector_max(n, in);
}
float ref_vector_sum(size_t n, const float * in) {
    return distributions::vector_sum(n, in);
}
void ref_vector_log(size_t n, float * io) { distributions::vector_log(n, io); }
void ref_vector_exp(size_t n, float * io) { distributions::vector_exp(n, io); }
void ref_vector_lgamma(size_t n, float * io) {
    distributions::vector_lgamma(n, io);
}
void ref_vector_shift(size_t n, float * io, float s) {
    distributions::vector_shift(n, io, s);
}
void ref_vector_scale(size_t n, float * io, float s) {
    distributions::vector_scale(n, io, s);
}

// ---- mixture.hpp:48-163 MixtureDriver --------------------------------------
void * ref_driver_new() { return new Driver(); }
void ref_driver_delete(void * p) { delete static_cast<Driver *>(p); }
void ref_driver_init(void * p, const int * counts, size_t n) {
    Driver * d = static_cast<Driver *>(p);
    d->counts().assign(counts, counts + n);
    d->init(NullModel());
}
int ref_driver_add_value(void * p, size_t groupid) {
    return static_cast<Driver *>(p)->add_value(NullModel(), groupid);
}
int ref_driver_remove_value(void * p, size_t groupid) {
    return static_cast<Driver *>(p)->remove_value(NullModel(), groupid);
}
size_t ref_driver_size(void * p) {
    return static_cast<Driver *>(p)->counts().size();
}
size_t ref_driver_sample_size(void * p) {
    return static_cast<Driver *>(p)->sample_size();
}
void ref_driver_counts(void * p, int * out) {
    Driver * d = static_cast<Driver *>(p);
    memcpy(out, d->counts().data(), d->counts().size() * sizeof(int));
}
size_t ref_driver_empty_count(void * p) {
    return static_cast<Driver *>(p)->empty_groupids().size();
}
int ref_driver_is_empty(void * p, size_t groupid) {
    Driver * d = static_cast<Driver *>(p);
    return d->empty_groupids().find(groupid) != d->empty_groupids().end();
}

// ---- mixture.hpp:460-521 MixtureIdTracker ----------------------------------
void * ref_tracker_new() { return new distributions::MixtureIdTracker(); }
void ref_tracker_delete(void * p) {
    delete static_cast<distributions::MixtureIdTracker *>(p);
}
void ref_tracker_init(void * p, size_t n) {
    static_cast<distributions::MixtureIdTracker *>(p)->init(n);
}
void ref_tracker_add_group(void * p) {
    static_cast<distributions::MixtureIdTracker *>(p)->add_group();
}
void ref_tracker_remove_group(void * p, uint32_t packed) {
    static_cast<distributions::MixtureIdTracker *>(p)->remove_group(packed);
}
uint32_t ref_tracker_packed_to_global(void * p, uint32_t packed) {
    return static_cast<distributions::MixtureIdTracker *>(p)
        ->packed_to_global(packed);
}
uint32_t ref_tracker_global_to_packed(void * p, uint32_t global) {
    return static_cast<distributions::MixtureIdTracker *>(p)
        ->global_to_packed(global);
}
size_t ref_tracker_packed_size(void * p) {
    return static_cast<distributions::MixtureIdTracker *>(p)->packed_size();
}

}  // extern "C"
