/* TEST INFRASTRUCTURE ONLY -- see oracle.h for scope and pinning status.
 *
 * Plain-C restatement of the reference hot path.  Compiled WITHOUT
 * -ffast-math and without FP contraction: every float operation below is
 * written in the order the reference's release build executes it (where that
 * could be observed in oracle/_ref) or in source order (where the reference
 * cannot be compiled here).  FTZ/DAZ is switched on explicitly because the
 * reference's -ffast-math link pulls in crtfastmath.o.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>

#include "ref_tables.h"

/* ------------------------------------------------------------------------ */
/* helpers                                                                  */

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

unsigned orc_ftz_enable(void) {
    unsigned saved = _mm_getcsr();
    _mm_setcsr(saved | 0x8040u); /* FTZ | DAZ, as crtfastmath.o does */
    return saved;
}
void orc_ftz_restore(unsigned saved) { _mm_setcsr(saved); }

__attribute__((constructor)) static void orc_ctor(void) { orc_ftz_enable(); }

#define LOGT(i) u2f(DIST_REF_LOG_TABLE[i])

/* ------------------------------------------------------------------------ */
/* special.hpp                                                              */

/* special.hpp:57-67 (FastLog::log, N=14), table special.cc:35-44 */
float orc_fast_log(float x) {
    int32_t intx = (int32_t)f2u(x);
    const int exp = ((intx >> 23) & 255) - 127;
    const int man = (intx & 0x7FFFFF) >> (23 - 14);
    return ((float)exp + LOGT(man)) * 0.69314718055994529f;
}

/* vendor/fmath.hpp:438-459 (fmath::exp, the SSE branch), special.hpp:87-89 */
float orc_fast_exp(float x) {
    const float a = u2f(DIST_REF_EXP_AB[0]); /* 1024 / logf(2) */
    const float b = u2f(DIST_REF_EXP_AB[1]); /* logf(2) / 1024 */
    __m128 x1 = _mm_set_ss(x);
    int limit = _mm_cvtss_si32(x1) & 0x7fffffff;
    if (limit > 0x42b00000) {
        x1 = _mm_min_ss(x1, _mm_set_ss(88.0f));
        x1 = _mm_max_ss(x1, _mm_set_ss(-88.0f));
    }
    int r = _mm_cvtss_si32(_mm_mul_ss(x1, _mm_set_ss(a)));
    unsigned v = (unsigned)r & 1023u;
    int u = r >> 10;
    uint32_t bits = ((uint32_t)(u + 127) << 23) | DIST_REF_EXP_TABLE[v];
    /* source: t = x - r*b; return (1 + t) * fi.  The release build
     * (-ffast-math) evaluates (x + 1) - r*b, observed in oracle/_ref for
     * vector_exp (vector_math.cc:190-221) and the shim loop. */
    return ((_mm_cvtss_f32(x1) + 1.0f) - (float)r * b) * u2f(bits);
}

/* special.hpp:114-171, coefficients special.cc:144-211 */
float orc_fast_lgamma(float y) {
    if (y < 2.5f || 4294967295.0f <= y) {
        return lgammaf(y); /* glibc libm, as the reference */
    }
    int32_t x = (int32_t)f2u(y);
    int c = (x >> 23) - 127; /* y >= 2.5: normal, positive */
    int pos = c * 6;
    float a5 = u2f(DIST_REF_LGAMMA_COEFF5[pos]);
    float a4 = u2f(DIST_REF_LGAMMA_COEFF5[pos + 1]);
    float a3 = u2f(DIST_REF_LGAMMA_COEFF5[pos + 2]);
    float a2 = u2f(DIST_REF_LGAMMA_COEFF5[pos + 3]);
    float a1 = u2f(DIST_REF_LGAMMA_COEFF5[pos + 4]);
    float a0 = u2f(DIST_REF_LGAMMA_COEFF5[pos + 5]);
    double yprod = y;
    double sum = a0;
    sum += a1 * yprod;
    yprod *= y;
    sum += a2 * yprod;
    yprod *= y;
    sum += a3 * yprod;
    yprod *= y;
    sum += a4 * yprod;
    yprod *= y;
    sum += a5 * yprod;
    return (float)sum;
}

/* special.hpp:208-214, table special.cc:213-230 */
float orc_fast_log_factorial(uint32_t n) {
    if (n < 64) {
        return u2f(DIST_REF_LOG_FACTORIAL[n]);
    }
    return orc_fast_lgamma((float)(n + 1u));
}

/* special.hpp:224-273, coefficients special.cc:232-269 */
float orc_fast_lgamma_nu(float nu) {
    if (nu < 0.0625f || 4294967295.0f <= nu) {
        /* source: lgammaf(nu*0.5f + 0.5f) - lgammaf(nu*0.5f); release build
         * forms the first argument as (nu + 1) * 0.5 */
        return lgammaf((nu + 1.0f) * 0.5f) - lgammaf(nu * 0.5f);
    }
    int32_t x = (int32_t)f2u(nu);
    int c = (x >> 23) - 127;
    int pos = ((c + 4) / 2) * 4;
    float a3 = u2f(DIST_REF_LGAMMA_NU_COEFF3[pos]);
    float a2 = u2f(DIST_REF_LGAMMA_NU_COEFF3[pos + 1]);
    float a1 = u2f(DIST_REF_LGAMMA_NU_COEFF3[pos + 2]);
    float a0 = u2f(DIST_REF_LGAMMA_NU_COEFF3[pos + 3]);
    /* source (special.hpp:234): a0 + x*a1 + x*x*a2 + x*x*x*a3.  The release
     * build (-ffast-math) evaluates (a0 + xx*a2) + x*(a1 + xx*a3), xx = x*x,
     * observed in oracle/_ref. */
    float xx = nu * nu;
    float p = xx * a3 + a1;
    float q = xx * a2 + a0;
    return q + p * nu;
}

void orc_vec_fast_log(size_t n, const float * in, float * out) {
    for (size_t i = 0; i < n; ++i) out[i] = orc_fast_log(in[i]);
}
void orc_vec_fast_exp(size_t n, const float * in, float * out) {
    for (size_t i = 0; i < n; ++i) out[i] = orc_fast_exp(in[i]);
}
void orc_vec_fast_lgamma(size_t n, const float * in, float * out) {
    for (size_t i = 0; i < n; ++i) out[i] = orc_fast_lgamma(in[i]);
}
void orc_vec_fast_lgamma_nu(size_t n, const float * in, float * out) {
    for (size_t i = 0; i < n; ++i) out[i] = orc_fast_lgamma_nu(in[i]);
}
void orc_vec_fast_log_factorial(size_t n, const uint32_t * in, float * out) {
    for (size_t i = 0; i < n; ++i) out[i] = orc_fast_log_factorial(in[i]);
}

/* special.cc:35-44 with scalar log2f (what the source says; the release
 * build calls libmvec instead, see gen_ref_tables.py) */
void orc_formula_log_table(float * out) {
    for (int i = 0; i < 16384; ++i) {
        float v = 1.0f + ((float)i * (float)(1 << 9)) / (float)(1 << 23);
        out[i] = log2f(v);
    }
}
/* vendor/fmath.hpp:165-172 with scalar powf */
void orc_formula_exp_table(uint32_t * out) {
    for (int i = 0; i < 1024; ++i) {
        float y = powf(2.0f, (float)i / 1024);
        out[i] = f2u(y) & 0x7fffffu;
    }
}

/* ------------------------------------------------------------------------ */
/* vector_math.cc (operation order as compiled: checked against _ref)       */

/* vector_math.cc:160-168; release build evaluates (io + in1) - in2 */
void orc_vector_add_subtract(size_t n, float * io, const float * a,
                             const float * b) {
    for (size_t i = 0; i < n; ++i) io[i] = (io[i] + a[i]) - b[i];
}
/* vector_math.cc:170-178 */
void orc_vector_add_subtract_scalar(size_t n, float * io, float a,
                                    const float * b) {
    for (size_t i = 0; i < n; ++i) io[i] = (io[i] + a) - b[i];
}
/* vector_math.cc:132-139 */
void orc_vector_add(size_t n, float * io, const float * a) {
    for (size_t i = 0; i < n; ++i) io[i] += a[i];
}
/* vector_math.cc:74-83 */
float orc_vector_max(size_t n, const float * in) {
    float res = in[0];
    for (size_t i = 0; i < n; ++i) {
        float x = in[i];
        res = x > res ? x : res;
    }
    return res;
}

/* ------------------------------------------------------------------------ */
/* random_fwd.hpp:34 rng_t = std::default_random_engine = minstd_rand0      */

#define ORC_M 2147483647ull
#define ORC_A 16807ull

/* libstdc++ linear_congruential_engine::seed */
uint32_t orc_rng_seed(uint64_t seed) {
    uint64_t s = seed % ORC_M;
    return (uint32_t)(s == 0 ? 1 : s);
}
uint32_t orc_rng_next(uint32_t * state) {
    *state = (uint32_t)((ORC_A * (uint64_t)*state) % ORC_M);
    return *state;
}
/* random.hpp:47-50: std::uniform_real_distribution<float>(0,1) over one
 * engine step = generate_canonical<float,24>: float(x - min) / 2^31, clamped
 * below 1 */
float orc_sample_unif01(uint32_t * state) {
    uint32_t x = orc_rng_next(state);
    float ret = (float)(uint64_t)(x - 1u) / 2147483648.0f;
    if (ret >= 1.0f) ret = u2f(0x3f7fffffu); /* nextafter(1, 0) */
    return ret;
}
/* state after `steps` further engine steps: state * a^steps mod m */
uint32_t orc_rng_jump(uint32_t state, uint64_t steps) {
    uint64_t result = state, base = ORC_A;
    while (steps) {
        if (steps & 1) result = (result * base) % ORC_M;
        base = (base * base) % ORC_M;
        steps >>= 1;
    }
    return (uint32_t)result;
}

/* random.cc:94-106 */
float orc_scores_to_likelihoods(size_t n, float * scores) {
    float max_score = orc_vector_max(n, scores);
    float total = 0;
    for (size_t i = 0; i < n; ++i) {
        total += scores[i] = orc_fast_exp(scores[i] - max_score);
    }
    return total;
}
static size_t sample_from_likelihoods_u(size_t n, const float * l,
                                        float total, float u) {
    float t = total * u;
    for (size_t i = 0; i < n; ++i) {
        t -= l[i];
        if (t <= 0) return i;
    }
    return n - 1;
}
/* random.hpp:316-333 */
size_t orc_sample_from_likelihoods(uint32_t * rng, size_t n, const float * l,
                                   float total) {
    return sample_from_likelihoods_u(n, l, total, orc_sample_unif01(rng));
}
/* random.hpp:361-366 */
size_t orc_sample_from_scores_overwrite(uint32_t * rng, size_t n,
                                        float * scores) {
    float total = orc_scores_to_likelihoods(n, scores);
    return orc_sample_from_likelihoods(rng, n, scores, total);
}
size_t orc_sample_from_scores_u(size_t n, float * scores, float u) {
    float total = orc_scores_to_likelihoods(n, scores);
    return sample_from_likelihoods_u(n, scores, total, u);
}
/* random.cc:77-92 */
float orc_log_sum_exp(size_t n, const float * scores) {
    if (n == 0) return 0.f;
    float max_score = orc_vector_max(n, scores);
    float total = 0;
    for (size_t i = 0; i < n; ++i) {
        total += orc_fast_exp(scores[i] - max_score);
    }
    return orc_fast_log(total) + max_score;
}
/* random.hpp:300-313 */
size_t orc_sample_discrete(uint32_t * rng, size_t dim, const float * probs) {
    float t = orc_sample_unif01(rng);
    for (size_t i = 0; i + 1 < dim; ++i) {
        t -= probs[i];
        if (t < 0) return i;
    }
    return dim - 1;
}

/* ------------------------------------------------------------------------ */
/* clustering.hpp:81-123                                                    */

float orc_py_score_add_value(float alpha, float d, int group_size,
                             int nonempty_group_count, int sample_size,
                             int empty_group_count) {
    if (group_size == 0) {
        float numer = alpha + d * (float)nonempty_group_count;
        float denom = ((float)sample_size + alpha) * (float)empty_group_count;
        return orc_fast_log(numer / denom);
    }
    return orc_fast_log(((float)group_size - d) / ((float)sample_size + alpha));
}
float orc_py_score_remove_value(float alpha, float d, int group_size,
                                int nonempty_group_count, int sample_size,
                                int empty_group_count) {
    group_size -= 1;
    if (group_size == 0) nonempty_group_count -= 1;
    sample_size -= 1;
    return -orc_py_score_add_value(alpha, d, group_size, nonempty_group_count,
                                   sample_size, empty_group_count);
}

/* ------------------------------------------------------------------------ */
/* feature slaves                                                           */

typedef struct {
    orc_shared sh;
    float * betas;   /* DPD: owned copy */
    float alpha_sum; /* DD: dd.hpp:403-406; DPD: alpha */
    int K, cap;
    /* suffstats, SoA over groups */
    int32_t * i0;    /* DD/DPD count_sum|total, BB heads, GP count, NICH count */
    int32_t * i1;    /* BB tails, GP sum */
    float * f0;      /* GP log_prod, NICH mean */
    float * f1;      /* NICH count_times_variance */
    int32_t * cnt;   /* DD/DPD counts[cap][dim] */
    /* value-scorer caches */
    float * c0;      /* DD/DPD shift, BB heads, GP score, NICH score */
    float * c1;      /* BB tails, GP post_alpha, NICH log_coeff */
    float * c2;      /* GP score_coeff, NICH precision */
    float * c3;      /* NICH mean */
    float * S;       /* DD/DPD scores_[dim][cap] */
} feat;

struct orc_mix {
    float alpha, d;
    int cluster;          /* 0: PitmanYor (cached), 1: LowEntropy (generic driver) */
    int dataset_size;     /* LowEntropy */
    int K, cap;
    int32_t * counts;
    float * shifted;
    int n_empty;
    int64_t sample_size;
    int F;
    feat * f;
    /* tracker */
    uint32_t * p2g;
    int p2g_size, p2g_cap;
    int32_t * g2p;
    uint32_t global_size;
    int g2p_cap;
};

static int is_cat(int kind) { return kind == ORC_DD || kind == ORC_DPD; }

static void feat_reserve(feat * f, int need) {
    if (need <= f->cap) return;
    int ncap = f->cap ? f->cap : 16;
    while (ncap < need) ncap *= 2;
    int dim = f->sh.dim;
    f->i0 = realloc(f->i0, sizeof(int32_t) * ncap);
    f->i1 = realloc(f->i1, sizeof(int32_t) * ncap);
    f->f0 = realloc(f->f0, sizeof(float) * ncap);
    f->f1 = realloc(f->f1, sizeof(float) * ncap);
    f->c0 = realloc(f->c0, sizeof(float) * ncap);
    f->c1 = realloc(f->c1, sizeof(float) * ncap);
    f->c2 = realloc(f->c2, sizeof(float) * ncap);
    f->c3 = realloc(f->c3, sizeof(float) * ncap);
    if (is_cat(f->sh.kind)) {
        f->cnt = realloc(f->cnt, sizeof(int32_t) * (size_t)ncap * dim);
        float * nS = malloc(sizeof(float) * (size_t)ncap * dim);
        for (int v = 0; v < dim && f->S; ++v)
            memcpy(nS + (size_t)v * ncap, f->S + (size_t)v * f->cap,
                   sizeof(float) * f->K);
        free(f->S);
        f->S = nS;
    }
    f->cap = ncap;
}

static float cat_prior(const feat * f, uint32_t v) {
    /* DD: alphas[v] (dd.hpp:376); DPD: alpha * betas[v] (dpd.hpp:424) */
    if (f->sh.kind == ORC_DD) return f->sh.alphas[v];
    return f->sh.p[0] * f->betas[v];
}

/* Group::init (dd.hpp:113-121, bb.hpp:95-100, gp.hpp:103-107,
 * nich.hpp:117-123, dpd.hpp:182-186) */
static void group_init(feat * f, int k) {
    f->i0[k] = 0;
    f->i1[k] = 0;
    f->f0[k] = 0.f;
    f->f1[k] = 0.f;
    if (is_cat(f->sh.kind))
        memset(f->cnt + (size_t)k * f->sh.dim, 0, sizeof(int32_t) * f->sh.dim);
}

/* Group::add_value (dd.hpp:123-130, bb.hpp:102-107, gp.hpp:109-116,
 * nich.hpp:125-133, dpd.hpp:188-195) */
static void group_add(feat * f, int k, uint32_t value) {
    switch (f->sh.kind) {
    case ORC_DD:
    case ORC_DPD:
        f->i0[k] += 1;
        f->cnt[(size_t)k * f->sh.dim + value] += 1;
        break;
    case ORC_BB:
        if (value) f->i0[k] += 1; else f->i1[k] += 1;
        break;
    case ORC_GP:
        f->i0[k] = (int32_t)((uint32_t)f->i0[k] + 1u);
        f->i1[k] = (int32_t)((uint32_t)f->i1[k] + value);
        f->f0[k] += orc_fast_log_factorial(value);
        break;
    case ORC_BNB:   /* bnb.hpp:106-112 */
        f->i0[k] = (int32_t)((uint32_t)f->i0[k] + 1u);
        f->i1[k] = (int32_t)((uint32_t)f->i1[k] + value);
        break;
    case ORC_NICH: {
        float x = u2f(value);
        f->i0[k] += 1;
        float delta = x - f->f0[k];
        f->f0[k] += delta / (float)f->i0[k];
        f->f1[k] += delta * (x - f->f0[k]);
        break;
    }
    }
}

/* Group::remove_value (dd.hpp:142-149, bb.hpp:117-122, gp.hpp:128-135,
 * nich.hpp:146-165, dpd.hpp:207-214) */
static void group_remove(feat * f, int k, uint32_t value) {
    switch (f->sh.kind) {
    case ORC_DD:
    case ORC_DPD:
        f->i0[k] -= 1;
        f->cnt[(size_t)k * f->sh.dim + value] -= 1;
        break;
    case ORC_BB:
        if (value) f->i0[k] -= 1; else f->i1[k] -= 1;
        break;
    case ORC_GP:
        f->i0[k] = (int32_t)((uint32_t)f->i0[k] - 1u);
        f->i1[k] = (int32_t)((uint32_t)f->i1[k] - value);
        f->f0[k] -= orc_fast_log_factorial(value);
        break;
    case ORC_BNB:   /* bnb.hpp:124-130 */
        f->i0[k] = (int32_t)((uint32_t)f->i0[k] - 1u);
        f->i1[k] = (int32_t)((uint32_t)f->i1[k] - value);
        break;
    case ORC_NICH: {
        float x = u2f(value);
        float total = f->f0[k] * (float)f->i0[k];
        float delta = x - f->f0[k];
        f->i0[k] -= 1;
        if (f->i0[k] == 0) {
            f->f0[k] = 0.f;
        } else {
            f->f0[k] = (total - x) / (float)f->i0[k];
        }
        if (f->i0[k] <= 1) {
            f->f1[k] = 0.f;
        } else {
            f->f1[k] -= delta * (x - f->f0[k]);
        }
        break;
    }
    }
}

/* the per-group cache entries as a pure function of the group's suffstats */
typedef struct { float c0, c1, c2, c3; } scorer4;

/* bb.hpp:189-197, gp.hpp:198-207 (+ :56-61), nich.hpp:239-250 (+ :58-69) */
static scorer4 scorer_init(const orc_shared * sh, int32_t i0, int32_t i1,
                           float f0, float f1) {
    scorer4 s = {0, 0, 0, 0};
    switch (sh->kind) {
    case ORC_BB: {
        float alpha = sh->p[0] + (float)i0;
        float beta = sh->p[1] + (float)i1;
        s.c0 = orc_fast_log(alpha / (alpha + beta));
        s.c1 = orc_fast_log(beta / (alpha + beta));
        break;
    }
    case ORC_GP: {
        float post_alpha = sh->p[0] + (float)(uint32_t)i1;
        float post_inv_beta = sh->p[1] + (float)(uint32_t)i0;
        float score_coeff = -orc_fast_log(1.f + post_inv_beta);
        s.c0 = -orc_fast_lgamma(post_alpha)
             + post_alpha * (orc_fast_log(post_inv_beta) + score_coeff);
        s.c1 = post_alpha;
        s.c2 = score_coeff;
        break;
    }
    case ORC_BNB: {   /* bnb.hpp:55-61 (plus_group), 200-215 (Scorer::init) */
        float r = sh->p[2];
        float post_alpha = sh->p[0] + r * (float)(uint32_t)i0;
        float post_beta = sh->p[1] + (float)(uint32_t)i1;
        float alpha = post_alpha + r;
        s.c0 = orc_fast_lgamma(post_alpha + post_beta)
             - orc_fast_lgamma(post_alpha)
             - orc_fast_lgamma(post_beta)
             + orc_fast_lgamma(alpha);
        s.c1 = post_beta;
        s.c2 = alpha;
        break;
    }
    case ORC_NICH: {
        float mu = sh->p[0], kappa = sh->p[1], sigmasq = sh->p[2],
              nu = sh->p[3];
        float count = (float)i0, mean = f0, ctv = f1;
        float mu_1 = mu - mean;
        float post_kappa = kappa + count;
        float post_mu = (kappa * mu + mean * count) / post_kappa;
        float post_nu = nu + count;
        float post_sigmasq = 1.f / post_nu * (
            nu * sigmasq + ctv + (count * kappa * mu_1 * mu_1) / post_kappa);
        float lambda = post_kappa / ((post_kappa + 1.f) * post_sigmasq);
        s.c0 = orc_fast_lgamma_nu(post_nu)
             + 0.5f * orc_fast_log(lambda / (3.14159265358979f * post_nu));
        s.c1 = -0.5f * post_nu - 0.5f;
        s.c2 = lambda / post_nu;
        s.c3 = post_mu;
        break;
    }
    }
    return s;
}

/* MixtureValueScorer::update_group (dd.hpp:369-379, bb.hpp:247-256,
 * gp.hpp:262-273, nich.hpp:312-324, dpd.hpp:414-428) */
static void cache_update_group(feat * f, int k) {
    if (is_cat(f->sh.kind)) {
        f->c0[k] = orc_fast_log(f->alpha_sum + (float)f->i0[k]);
        for (int v = 0; v < f->sh.dim; ++v) {
            f->S[(size_t)v * f->cap + k] = orc_fast_log(
                cat_prior(f, v) + (float)f->cnt[(size_t)k * f->sh.dim + v]);
        }
    } else {
        scorer4 s = scorer_init(&f->sh, f->i0[k], f->i1[k], f->f0[k], f->f1[k]);
        f->c0[k] = s.c0; f->c1[k] = s.c1; f->c2[k] = s.c2; f->c3[k] = s.c3;
    }
}

/* MixtureValueScorer::add_value/remove_value: dd.hpp:381-397,458-467 (only
 * the touched value's entry and the shift); others = update_group */
static void cache_update_value(feat * f, int k, uint32_t value) {
    if (is_cat(f->sh.kind)) {
        f->S[(size_t)value * f->cap + k] = orc_fast_log(
            cat_prior(f, value)
            + (float)f->cnt[(size_t)k * f->sh.dim + value]);
        f->c0[k] = orc_fast_log(f->alpha_sum + (float)f->i0[k]);
    } else {
        cache_update_group(f, k);
    }
}

/* MixtureSlave::init = resize + update_all (mixture.hpp:354-359;
 * dd.hpp:399-421, bb.hpp:276-291, gp.hpp/nich.hpp update_all) */
static void cache_update_all(feat * f) {
    if (f->sh.kind == ORC_DD) {
        f->alpha_sum = 0;
        for (int v = 0; v < f->sh.dim; ++v) f->alpha_sum += f->sh.alphas[v];
    } else if (f->sh.kind == ORC_DPD) {
        f->alpha_sum = f->sh.p[0];
    }
    for (int k = 0; k < f->K; ++k) cache_update_group(f, k);
}

static void feat_add_group(feat * f) {   /* mixture.hpp:361-368 */
    feat_reserve(f, f->K + 1);
    int k = f->K++;
    group_init(f, k);
    cache_update_group(f, k);
}

static void feat_remove_group(feat * f, int k) {   /* mixture.hpp:370-375 */
    int last = f->K - 1;
    if (k != last) {
        f->i0[k] = f->i0[last]; f->i1[k] = f->i1[last];
        f->f0[k] = f->f0[last]; f->f1[k] = f->f1[last];
        f->c0[k] = f->c0[last]; f->c1[k] = f->c1[last];
        f->c2[k] = f->c2[last]; f->c3[k] = f->c3[last];
        if (is_cat(f->sh.kind)) {
            int dim = f->sh.dim;
            memcpy(f->cnt + (size_t)k * dim, f->cnt + (size_t)last * dim,
                   sizeof(int32_t) * dim);
            for (int v = 0; v < dim; ++v)
                f->S[(size_t)v * f->cap + k] = f->S[(size_t)v * f->cap + last];
        }
    }
    f->K = last;
}

/* score of `value` under a group given its four cache entries
 * (gp.hpp:209-217 / gp.cc:57-66, nich.hpp:252-259 / nich.cc:60-66,
 * bb.hpp:199-204) -- the term that is added to the accumulator */
static inline float noncat_term(int kind, scorer4 s, uint32_t value,
                                float log_factorial_value) {
    switch (kind) {
    case ORC_BB:
        return value ? s.c0 : s.c1;
    case ORC_GP: {
        float fv = (float)value;
        return s.c0 + orc_fast_lgamma(s.c1 + fv) - log_factorial_value
             + s.c2 * fv;
    }
    case ORC_BNB: {   /* bnb.hpp:217-223, 316-327 */
        float beta = s.c1 + (float)value;
        return s.c0 + orc_fast_lgamma(beta) - orc_fast_lgamma(beta + s.c2);
    }
    default: { /* ORC_NICH */
        float x = u2f(value);
        float d = x - s.c3;
        float temp = 1.f + s.c2 * (d * d);
        return s.c0 + s.c1 * orc_fast_log(temp);
    }
    }
}

/* row-score of a categorical value: which table row, or the scalar fallback
 * of dpd.hpp:534-542 for OTHER */
static inline float cat_entry(const feat * f, uint32_t value, int k) {
    if (f->sh.kind == ORC_DPD && value == 0xFFFFFFFFu)
        return orc_fast_log(f->sh.p[0] * f->sh.p[1]);
    return f->S[(size_t)value * f->cap + k];
}

/* MixtureValueScorer::score_value (dd.hpp:433-445 -> vector_add_subtract,
 * bb.hpp:303-313 -> vector_add, gp.cc:32-67, nich.cc:33-66, dpd.hpp:517-543) */
static void feat_score_value(const feat * f, uint32_t value, float * acc) {
    if (is_cat(f->sh.kind)) {
        for (int k = 0; k < f->K; ++k)
            acc[k] = (acc[k] + cat_entry(f, value, k)) - f->c0[k];
        return;
    }
    float lf = f->sh.kind == ORC_GP ? orc_fast_log_factorial(value) : 0.f;
    for (int k = 0; k < f->K; ++k) {
        scorer4 s = {f->c0[k], f->c1[k], f->c2[k], f->c3[k]};
        acc[k] += noncat_term(f->sh.kind, s, value, lf);
    }
}

/* MixtureValueScorer::score_value_group (dd.hpp:423-431, bb.hpp:293-301,
 * gp.hpp:300-310, nich.hpp:351-360, dpd.hpp:499-515) */
static float feat_score_value_group(const feat * f, int k, uint32_t value) {
    if (is_cat(f->sh.kind)) return cat_entry(f, value, k) - f->c0[k];
    float lf = f->sh.kind == ORC_GP ? orc_fast_log_factorial(value) : 0.f;
    scorer4 s = {f->c0[k], f->c1[k], f->c2[k], f->c3[k]};
    return noncat_term(f->sh.kind, s, value, lf);
}

/* ------------------------------------------------------------------------ */
/* vector_sum (vector_math.cc:85-93) as the release build executes it: the
 * -ffast-math loop is vectorised into four lane accumulators (element i goes
 * to lane i mod 4) over the first 4*floor(n/4) elements, combined as
 * (lane1 + lane3) + (lane0 + lane2), and the tail is added in order; fewer
 * than four elements are summed in order.  Pinned against oracle/_ref. */
float orc_vector_sum(size_t n, const float * x) {
    if (n < 4) {
        float s = 0.f;
        for (size_t i = 0; i < n; ++i) s += x[i];
        return s;
    }
    float lane[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t body = n & ~(size_t)3;
    for (size_t i = 0; i < body; i += 4)
        for (int j = 0; j < 4; ++j) lane[j] += x[i + j];
    float s = (lane[1] + lane[3]) + (lane[0] + lane[2]);
    for (size_t i = body; i < n; ++i) s += x[i];
    return s;
}

/* Clustering<int>::LowEntropy (clustering.hpp:245-331, clustering.cc:186-283) */
#include "le_table.h"

/* clustering.hpp:318-327 */
static float le_postpred_correction(float sample_size, int dataset_size) {
    float exponent = 0.45f - 0.1f / sample_size - 0.1f / (float)dataset_size;
    float scale = (float)dataset_size / sample_size;
    return orc_fast_log(scale) * exponent;
}
/* clustering.hpp:267-292 (nonempty_group_count is unused by the model) */
float orc_le_score_add_value(int dataset_size, int group_size,
                             int nonempty_group_count, int sample_size,
                             int empty_group_count) {
    (void)nonempty_group_count;
    if (group_size == 0) {
        float score = -orc_fast_log((float)empty_group_count);
        if (sample_size + 1 < dataset_size)
            score += le_postpred_correction((float)(sample_size + 1),
                                            dataset_size);
        return score;
    }
    const int very_large = 10000;
    float bigger = 1.f + (float)group_size;
    if (group_size > very_large) return 1.f + orc_fast_log(bigger);
    return orc_fast_log(bigger / (float)group_size) * (float)group_size
         + orc_fast_log(bigger);
}
/* clustering.hpp:294-309 */
float orc_le_score_remove_value(int dataset_size, int group_size,
                                int nonempty_group_count, int sample_size,
                                int empty_group_count) {
    return -orc_le_score_add_value(dataset_size, group_size - 1,
                                   nonempty_group_count, sample_size,
                                   empty_group_count);
}
/* clustering.cc:204-215 */
float orc_le_log_partition_function(int n) {
    if (n < 48) return u2f(DIST_LE_LOG_PARTITION[n]);
    float coeff = 0.28269584f;
    float log_z_max = (float)n * orc_fast_log((float)n);
    return log_z_max * (1.f + coeff * powf((float)n, -0.75f));
}
/* clustering.cc:221-248 */
float orc_le_score_counts(int dataset_size, const int * counts, size_t size) {
    unsigned saved = orc_ftz_enable();
    float score = 0.f;
    int sample_size = 0;
    for (size_t i = 0; i < size; ++i) {
        sample_size += counts[i];
        if (counts[i] > 1)
            score += (float)counts[i] * orc_fast_log((float)counts[i]);
    }
    if (sample_size != dataset_size) {
        float log_factor = le_postpred_correction((float)sample_size,
                                                  dataset_size);
        score += log_factor * (float)(size - 1);
        float n = orc_fast_log((float)sample_size);
        float N = orc_fast_log((float)dataset_size);
        score += 0.061f * n * (n - N) * powf(n + N, 0.75f);
    }
    score -= orc_le_log_partition_function(sample_size);
    orc_ftz_restore(saved);
    return score;
}
/* LowEntropy::sample_assignments (clustering.cc:250-283); the two-argument
 * sample_from_likelihoods takes its total from vector_sum (random.hpp:335-341);
 * one engine step per row */
void orc_le_sample_assignments(int dataset_size, int sample_size,
                               uint32_t * rng_state, int * assignments) {
    unsigned saved = orc_ftz_enable();
    int * counts = malloc(sizeof(int) * (size_t)(sample_size + 2));
    float * likelihoods = malloc(sizeof(float) * (size_t)(sample_size + 2));
    int n = 0;
    int size = 0;
    for (int i = 0; i < sample_size; ++i) {
        float likelihood_empty = orc_fast_exp(
            orc_le_score_add_value(dataset_size, 0, 0, size, 1));
        if (n == 0 || counts[n - 1]) {
            counts[n] = 0;
            likelihoods[n] = likelihood_empty;
            n += 1;
        } else {
            likelihoods[n - 1] = likelihood_empty;
        }
        float total = orc_vector_sum((size_t)n, likelihoods);
        int assign = (int)orc_sample_from_likelihoods(rng_state, (size_t)n,
                                                      likelihoods, total);
        assignments[i] = assign;
        counts[assign] += 1;
        size += 1;
        likelihoods[assign] = orc_fast_exp(
            orc_le_score_add_value(dataset_size, counts[assign], 0, 0, 1));
    }
    free(counts); free(likelihoods);
    orc_ftz_restore(saved);
}

/* ------------------------------------------------------------------------ */
/* driver: clustering.hpp:126-234 over mixture.hpp:48-163                   */

static void py_reserve(orc_mix * m, int need) {
    if (need <= m->cap) return;
    int ncap = m->cap ? m->cap : 16;
    while (ncap < need) ncap *= 2;
    m->counts = realloc(m->counts, sizeof(int32_t) * ncap);
    m->shifted = realloc(m->shifted, sizeof(float) * ncap);
    m->cap = ncap;
}
/* clustering.hpp:215-219 */
static void py_update_nonempty(orc_mix * m, int k) {
    if (m->cluster) return;   /* the generic driver keeps no cache */
    m->shifted[k] = orc_fast_log((float)m->counts[k] - m->d);
}
static float py_empty_score(float alpha, float d, int nonempty, int empty) {
    float numer = alpha + d * (float)(uint64_t)nonempty;
    float denom = (float)(uint64_t)empty;
    return orc_fast_log(numer / denom);
}
/* clustering.hpp:221-230 */
static void py_update_empty(orc_mix * m) {
    if (m->cluster) return;
    float s = py_empty_score(m->alpha, m->d, m->K - m->n_empty, m->n_empty);
    for (int k = 0; k < m->K; ++k)
        if (m->counts[k] == 0) m->shifted[k] = s;
}

void orc_mix_driver_init(orc_mix * m, const int * counts, int group_count) {
    py_reserve(m, group_count);
    m->K = group_count;
    m->sample_size = 0;
    m->n_empty = 0;
    for (int k = 0; k < group_count; ++k) {
        m->counts[k] = counts[k];
        m->sample_size += counts[k];
        if (counts[k] == 0) m->n_empty += 1;
    }
    for (int k = 0; k < group_count; ++k)
        if (m->counts[k]) py_update_nonempty(m, k);
    py_update_empty(m);
}

/* clustering.hpp:163-176 + mixture.hpp:73-92 */
int orc_mix_driver_add_value(orc_mix * m, int g) {
    int add_group = (m->counts[g] == 0);
    m->counts[g] += 1;
    m->sample_size += 1;
    if (add_group) {
        py_reserve(m, m->K + 1);
        m->counts[m->K] = 0;   /* the filled group leaves the empty set, */
        m->K += 1;             /* a fresh empty group joins it at the end */
        py_update_empty(m);
    }
    py_update_nonempty(m, g);
    return add_group;
}

/* clustering.hpp:178-193 + mixture.hpp:94-122 */
int orc_mix_driver_remove_value(orc_mix * m, int g) {
    m->counts[g] -= 1;
    m->sample_size -= 1;
    int remove_group = (m->counts[g] == 0);
    if (remove_group) {
        int last = m->K - 1;
        if (g != last) {
            m->counts[g] = m->counts[last];
            m->shifted[g] = m->shifted[last];
        }
        m->K = last;
        py_update_empty(m);
    } else {
        py_update_nonempty(m, g);
    }
    return remove_group;
}

/* clustering.hpp:195-208 */
void orc_mix_driver_score_value(const orc_mix * m, float * scores) {
    if (m->cluster) {   /* MixtureDriver::score_value, mixture.hpp:124-141 */
        for (int k = 0; k < m->K; ++k)
            scores[k] = orc_le_score_add_value(
                m->dataset_size, m->counts[k], m->K - m->n_empty,
                (int)m->sample_size, m->n_empty);
        return;
    }
    const float shift =
        -orc_fast_log((float)(uint64_t)m->sample_size + m->alpha);
    for (int k = 0; k < m->K; ++k) scores[k] = m->shifted[k] + shift;
}

int orc_mix_size(const orc_mix * m) { return m->K; }
int orc_mix_sample_size(const orc_mix * m) { return (int)m->sample_size; }
int orc_mix_empty_count(const orc_mix * m) { return m->n_empty; }
void orc_mix_get_counts(const orc_mix * m, int * out) {
    memcpy(out, m->counts, sizeof(int) * m->K);
}
void orc_mix_get_shifted(const orc_mix * m, float * out) {
    memcpy(out, m->shifted, sizeof(float) * m->K);
}

/* ------------------------------------------------------------------------ */
/* id tracker: mixture.hpp:460-521                                          */

void orc_mix_tracker_add_group(orc_mix * m) {
    if (m->p2g_size + 1 > m->p2g_cap) {
        m->p2g_cap = m->p2g_cap ? 2 * m->p2g_cap : 16;
        m->p2g = realloc(m->p2g, sizeof(uint32_t) * m->p2g_cap);
    }
    if ((int)m->global_size + 1 > m->g2p_cap) {
        m->g2p_cap = m->g2p_cap ? 2 * m->g2p_cap : 16;
        m->g2p = realloc(m->g2p, sizeof(int32_t) * m->g2p_cap);
    }
    uint32_t packed = (uint32_t)m->p2g_size;
    uint32_t global = m->global_size++;
    m->p2g[m->p2g_size++] = global;
    m->g2p[global] = (int32_t)packed;
}
void orc_mix_tracker_init(orc_mix * m, int group_count) {
    m->p2g_size = 0;
    m->global_size = 0;
    for (int i = 0; i < group_count; ++i) orc_mix_tracker_add_group(m);
}
void orc_mix_tracker_remove_group(orc_mix * m, uint32_t packed) {
    uint32_t global = m->p2g[packed];
    m->g2p[global] = -1;
    m->p2g[packed] = m->p2g[m->p2g_size - 1];
    m->p2g_size -= 1;
    if ((int)packed != m->p2g_size) {
        m->g2p[m->p2g[packed]] = (int32_t)packed;
    }
}
uint32_t orc_mix_global_size(const orc_mix * m) { return m->global_size; }
uint32_t orc_mix_packed_to_global(const orc_mix * m, uint32_t packed) {
    return m->p2g[packed];
}
uint32_t orc_mix_global_to_packed(const orc_mix * m, uint32_t global) {
    return (uint32_t)m->g2p[global];
}

/* ------------------------------------------------------------------------ */
/* construction / slave API                                                 */

orc_mix * orc_mix_create(float alpha, float d, int F,
                         const orc_shared * shareds) {
    orc_mix * m = calloc(1, sizeof(orc_mix));
    m->alpha = alpha;
    m->d = d;
    m->cluster = 0;
    m->dataset_size = 0;
    m->F = F;
    m->f = calloc(F > 0 ? F : 1, sizeof(feat));
    for (int i = 0; i < F; ++i) {
        m->f[i].sh = shareds[i];
        if (shareds[i].kind == ORC_DPD) {
            m->f[i].betas = malloc(sizeof(float) * shareds[i].dim);
            memcpy(m->f[i].betas, shareds[i].betas,
                   sizeof(float) * shareds[i].dim);
            m->f[i].sh.betas = m->f[i].betas;
        }
    }
    return m;
}
void orc_mix_set_low_entropy(orc_mix * m, int dataset_size) {
    m->cluster = 1;
    m->dataset_size = dataset_size;
}
void orc_mix_destroy(orc_mix * m) {
    if (!m) return;
    for (int i = 0; i < m->F; ++i) {
        feat * f = &m->f[i];
        free(f->i0); free(f->i1); free(f->f0); free(f->f1); free(f->cnt);
        free(f->c0); free(f->c1); free(f->c2); free(f->c3); free(f->S);
        free(f->betas);
    }
    free(m->f); free(m->counts); free(m->shifted); free(m->p2g); free(m->g2p);
    free(m);
}

void orc_mix_slave_clear(orc_mix * m, int fi) { m->f[fi].K = 0; }
void orc_mix_slave_append_empty(orc_mix * m, int fi) {
    feat * f = &m->f[fi];
    feat_reserve(f, f->K + 1);
    group_init(f, f->K);
    f->K += 1;
}
void orc_mix_slave_group_add_value(orc_mix * m, int fi, int g, uint32_t v) {
    group_add(&m->f[fi], g, v);
}
void orc_mix_slave_init(orc_mix * m, int fi) { cache_update_all(&m->f[fi]); }
void orc_mix_slave_add_group(orc_mix * m, int fi) { feat_add_group(&m->f[fi]); }
void orc_mix_slave_remove_group(orc_mix * m, int fi, int g) {
    feat_remove_group(&m->f[fi], g);
}
/* mixture.hpp:377-384 */
void orc_mix_slave_add_value(orc_mix * m, int fi, int g, uint32_t v) {
    group_add(&m->f[fi], g, v);
    cache_update_value(&m->f[fi], g, v);
}
/* mixture.hpp:386-398 */
void orc_mix_slave_remove_value(orc_mix * m, int fi, int g, uint32_t v) {
    group_remove(&m->f[fi], g, v);
    cache_update_value(&m->f[fi], g, v);
}
float orc_mix_slave_score_value_group(const orc_mix * m, int fi, int g,
                                      uint32_t v) {
    return feat_score_value_group(&m->f[fi], g, v);
}
void orc_mix_slave_score_value(const orc_mix * m, int fi, uint32_t v,
                               float * acc) {
    feat_score_value(&m->f[fi], v, acc);
}
int orc_mix_slave_size(const orc_mix * m, int fi) { return m->f[fi].K; }
void orc_mix_slave_get_group(const orc_mix * m, int fi, int g,
                             uint32_t * out) {
    const feat * f = &m->f[fi];
    switch (f->sh.kind) {
    case ORC_DD:
    case ORC_DPD:
        out[0] = (uint32_t)f->i0[g];
        memcpy(out + 1, f->cnt + (size_t)g * f->sh.dim, 4 * f->sh.dim);
        break;
    case ORC_BB:
    case ORC_BNB:
        out[0] = (uint32_t)f->i0[g]; out[1] = (uint32_t)f->i1[g];
        break;
    case ORC_GP:
        out[0] = (uint32_t)f->i0[g]; out[1] = (uint32_t)f->i1[g];
        out[2] = f2u(f->f0[g]);
        break;
    case ORC_NICH:
        out[0] = (uint32_t)f->i0[g]; out[1] = f2u(f->f0[g]);
        out[2] = f2u(f->f1[g]);
        break;
    }
}

/* Group::score_value = Scorer::init + Scorer::eval (dd.hpp:222-245,
 * bb.hpp:185-205, gp.hpp:194-217, nich.hpp:232-259); `group` laid out as
 * orc_mix_slave_get_group writes it */
float orc_group_score_value(const orc_shared * sh, const uint32_t * group,
                            uint32_t value) {
    if (sh->kind == ORC_DD) {
        float alpha_sum = 0, mine = 0;
        for (int v = 0; v < sh->dim; ++v) {
            float alpha = sh->alphas[v] + (float)(int32_t)group[1 + v];
            if ((uint32_t)v == value) mine = alpha;
            alpha_sum += alpha;
        }
        return orc_fast_log(mine / alpha_sum);
    }
    if (sh->kind == ORC_DPD) { /* dpd.hpp:223-232 */
        float alpha = sh->p[0];
        float numer = value == 0xFFFFFFFFu
            ? alpha * sh->p[1]
            : alpha * sh->betas[value] + (float)(int32_t)group[1 + value];
        float denom = alpha + (float)(int32_t)group[0];
        return orc_fast_log(numer / denom);
    }
    scorer4 s;
    if (sh->kind == ORC_NICH)
        s = scorer_init(sh, (int32_t)group[0], 0, u2f(group[1]), u2f(group[2]));
    else
        s = scorer_init(sh, (int32_t)group[0], (int32_t)group[1], 0.f, 0.f);
    float lf = sh->kind == ORC_GP ? orc_fast_log_factorial(value) : 0.f;
    return noncat_term(sh->kind, s, value, lf);
}

/* PitmanYor::sample_assignments (src/clustering.cc:67-142): sequential CRP
 * draw over a growing likelihood vector; one engine step per row after the
 * first */
void orc_py_sample_assignments(float alpha, float d, int size,
                               uint32_t * rng_state, int * assignments) {
    unsigned saved = orc_ftz_enable();
    float * likelihoods = malloc(sizeof(float) * (size_t)(size + 2));
    int n_like = 0;
    int table_count = 0;
    const float py_likelihood_new = 1 - d;
    likelihoods[n_like++] = alpha;
    if (size) {
        assignments[0] = 0;
        table_count = 1;
        likelihoods[n_like++] = alpha + d * table_count;
        likelihoods[0] = py_likelihood_new;
    }
    for (int i = 1; i < size; ++i) {
        float total = i + alpha;
        int assign = (int)orc_sample_from_likelihoods(rng_state, n_like,
                                                      likelihoods, total);
        assignments[i] = assign;
        if (assign == table_count) {
            table_count += 1;
            likelihoods[n_like++] = alpha + d * table_count;
            likelihoods[assign] = py_likelihood_new;
        } else {
            likelihoods[assign] += 1.0f;
        }
    }
    free(likelihoods);
    orc_ftz_restore(saved);
}

/* ------------------------------------------------------------------------ */
/* score_data                                            */

/* Group::score_data (dd.hpp:160-177, bb.hpp:141-151, gp.hpp:155-164,
 * nich.hpp:190-202, dpd.hpp:234-250); `group` as orc_mix_slave_get_group */
float orc_group_score_data(const orc_shared * sh, const uint32_t * group) {
    unsigned saved = orc_ftz_enable();
    float score = 0;
    if (sh->kind == ORC_DD || sh->kind == ORC_DPD) {
        float alpha_sum = 0;
        if (sh->kind == ORC_DPD) alpha_sum = sh->p[0];
        for (int v = 0; v < sh->dim; ++v) {
            float alpha = sh->kind == ORC_DD ? sh->alphas[v]
                                             : sh->p[0] * sh->betas[v];
            if (sh->kind == ORC_DD) alpha_sum += alpha;
            int32_t c = (int32_t)group[1 + v];
            if (sh->kind == ORC_DPD && c == 0) continue; /* sparse counter */
            score += orc_fast_lgamma(alpha + (float)c) - orc_fast_lgamma(alpha);
        }
        score += orc_fast_lgamma(alpha_sum)
               - orc_fast_lgamma(alpha_sum + (float)(int32_t)group[0]);
    } else if (sh->kind == ORC_BB) {
        float alpha = sh->p[0] + (float)(int32_t)group[0];
        float beta = sh->p[1] + (float)(int32_t)group[1];
        score += orc_fast_lgamma(alpha) - orc_fast_lgamma(sh->p[0]);
        score += orc_fast_lgamma(beta) - orc_fast_lgamma(sh->p[1]);
        score += orc_fast_lgamma(sh->p[0] + sh->p[1])
               - orc_fast_lgamma(alpha + beta);
    } else if (sh->kind == ORC_BNB) {   /* bnb.hpp:157-166 */
        float pa = sh->p[0] + sh->p[2] * (float)group[0];
        float pb = sh->p[1] + (float)group[1];
        score = orc_fast_lgamma(sh->p[0] + sh->p[1]) - orc_fast_lgamma(pa + pb);
        score += orc_fast_lgamma(pa) - orc_fast_lgamma(sh->p[0]);
        score += orc_fast_lgamma(pb) - orc_fast_lgamma(sh->p[1]);
    } else if (sh->kind == ORC_GP) {
        float post_alpha = sh->p[0] + (float)group[1];
        float post_inv_beta = sh->p[1] + (float)group[0];
        score = orc_fast_lgamma(post_alpha) - orc_fast_lgamma(sh->p[0]);
        score += sh->p[0] * orc_fast_log(sh->p[1])
               - post_alpha * orc_fast_log(post_inv_beta);
        score += -u2f(group[2]);
    } else {
        float mu = sh->p[0], kappa = sh->p[1], sigmasq = sh->p[2],
              nu = sh->p[3];
        float count = (float)(int32_t)group[0], mean = u2f(group[1]),
              ctv = u2f(group[2]);
        float mu_1 = mu - mean;
        float pk = kappa + count;
        float pnu = nu + count;
        float psig = 1.f / pnu * (
            nu * sigmasq + ctv + (count * kappa * mu_1 * mu_1) / pk);
        float log_pi = 1.1447298858493991f;
        score = orc_fast_lgamma(0.5f * pnu) - orc_fast_lgamma(0.5f * nu);
        score += 0.5f * orc_fast_log(kappa / pk);
        score += 0.5f * nu * (orc_fast_log(nu * sigmasq))
               - 0.5f * pnu * orc_fast_log(pnu * psig);
        score += -0.5f * count * log_pi;
    }
    orc_ftz_restore(saved);
    return score;
}

/* MixtureDataScorer::score_data (dd.hpp:250-256,287-318; bb.hpp:207-229;
 * gp.hpp:220-241; nich.hpp:262-288; dpd.hpp:344-374), float accumulation in
 * the reference's loop order (DD's final vector_sum as orc_vector_sum) */
static float slave_score_data_with(const feat * f, const orc_shared * sh,
                                   const float * betas);
float orc_mix_slave_score_data(const orc_mix * m, int fi) {
    const feat * f = &m->f[fi];
    return slave_score_data_with(f, &f->sh, f->betas);
}
static float slave_score_data_with(const feat * f, const orc_shared * sh,
                                   const float * betas) {
    unsigned saved = orc_ftz_enable();
    float result = 0;
    if (sh->kind == ORC_DD) {
        int dim = sh->dim;
        float * scores = calloc(dim + 1, sizeof(float));
        float * shared_part = calloc(dim + 1, sizeof(float));
        float alpha_sum = 0;
        for (int i = 0; i < dim; ++i) {
            alpha_sum += sh->alphas[i];
            shared_part[i] = orc_fast_lgamma(sh->alphas[i]);
        }
        shared_part[dim] = orc_fast_lgamma(alpha_sum);
        for (int k = 0; k < f->K; ++k) {
            if (!f->i0[k]) continue;
            for (int i = 0; i < dim; ++i)
                scores[i] += orc_fast_lgamma(
                    sh->alphas[i] + (float)f->cnt[(size_t)k * dim + i])
                           - shared_part[i];
            scores[dim] += shared_part[dim]
                         - orc_fast_lgamma(alpha_sum + (float)f->i0[k]);
        }
        result = orc_vector_sum((size_t)dim + 1, scores);   /* _eval */
        free(scores); free(shared_part);
    } else if (sh->kind == ORC_DPD) {
        float alpha = sh->p[0];
        float shared_total = orc_fast_lgamma(alpha);
        for (int k = 0; k < f->K; ++k) {
            if (!f->i0[k]) continue;
            for (int v = 0; v < sh->dim; ++v) {
                int32_t c = f->cnt[(size_t)k * sh->dim + v];
                if (!c) continue;
                float prior_i = betas[v] * alpha;
                result += orc_fast_lgamma(prior_i + (float)c)
                        - orc_fast_lgamma(alpha * betas[v]);
            }
            result += shared_total - orc_fast_lgamma(alpha + (float)f->i0[k]);
        }
    } else if (sh->kind == ORC_BB) {
        float shared_part = + orc_fast_lgamma(sh->p[0] + sh->p[1])
                            - orc_fast_lgamma(sh->p[0])
                            - orc_fast_lgamma(sh->p[1]);
        for (int k = 0; k < f->K; ++k) {
            float alpha = sh->p[0] + (float)f->i0[k];
            float beta = sh->p[1] + (float)f->i1[k];
            float group_part = + orc_fast_lgamma(alpha) + orc_fast_lgamma(beta)
                               - orc_fast_lgamma(alpha + beta);
            result += shared_part + group_part;
        }
    } else if (sh->kind == ORC_BNB) {   /* bnb.hpp:226-245 */
        float shared_part = orc_fast_lgamma(sh->p[0] + sh->p[1])
                          - orc_fast_lgamma(sh->p[0])
                          - orc_fast_lgamma(sh->p[1]);
        for (int k = 0; k < f->K; ++k) {
            if (!f->i0[k]) continue;
            float pa = sh->p[0] + sh->p[2] * (float)(uint32_t)f->i0[k];
            float pb = sh->p[1] + (float)(uint32_t)f->i1[k];
            result += orc_fast_lgamma(pa) + orc_fast_lgamma(pb)
                    - orc_fast_lgamma(pa + pb);
            result += shared_part;
        }
    } else if (sh->kind == ORC_GP) {
        float alpha_part = orc_fast_lgamma(sh->p[0]);
        float beta_part = sh->p[0] * orc_fast_log(sh->p[1]);
        for (int k = 0; k < f->K; ++k) {
            if (!f->i0[k]) continue;
            float post_alpha = sh->p[0] + (float)(uint32_t)f->i1[k];
            float post_inv_beta = sh->p[1] + (float)(uint32_t)f->i0[k];
            result += orc_fast_lgamma(post_alpha) - alpha_part;
            result += beta_part - post_alpha * orc_fast_log(post_inv_beta);
            result += -f->f0[k];
        }
    } else {
        float kappa = sh->p[1], sigmasq = sh->p[2], nu = sh->p[3],
              mu = sh->p[0];
        float nu_part = orc_fast_lgamma(0.5f * nu);
        float kappa_part = 0.5f * orc_fast_log(kappa);
        float sigmasq_part = 0.5f * nu * orc_fast_log(nu * sigmasq);
        float log_pi = 1.1447298858493991f;
        for (int k = 0; k < f->K; ++k) {
            if (!f->i0[k]) continue;
            float count = (float)f->i0[k], mean = f->f0[k], ctv = f->f1[k];
            float mu_1 = mu - mean;
            float pk = kappa + count;
            float pnu = nu + count;
            float psig = 1.f / pnu * (
                nu * sigmasq + ctv + (count * kappa * mu_1 * mu_1) / pk);
            result += orc_fast_lgamma(0.5f * pnu) - nu_part;
            result += kappa_part - 0.5f * orc_fast_log(pk);
            result += sigmasq_part - 0.5f * pnu * orc_fast_log(pnu * psig);
            result += -0.5f * log_pi * (float)f->i0[k];
        }
    }
    orc_ftz_restore(saved);
    return result;
}

/* score_data_grid (mixture.hpp:238-247: one score_data per candidate Shared;
 * DirichletDiscrete's incremental form, dd.hpp:259-345: _init on the first
 * candidate, then per candidate only the changed alphas are re-accumulated,
 * alpha_sum carried in binary64; _eval through orc_vector_sum) */
void orc_mix_slave_score_data_grid(const orc_mix * m, int fi,
                                   const orc_shared * shareds, size_t n,
                                   float * scores_out) {
    const feat * f = &m->f[fi];
    if (f->sh.kind != ORC_DD) {
        for (size_t i = 0; i < n; ++i)
            scores_out[i] = slave_score_data_with(f, &shareds[i],
                                                  shareds[i].betas);
        return;
    }
    if (!n) return;
    unsigned saved = orc_ftz_enable();
    const int dim = f->sh.dim;
    float * scores = calloc(dim + 1, sizeof(float));
    float * shared_part = calloc(dim + 1, sizeof(float));
    /* _init(shareds[0]) */
    float alpha_sum_f = 0;
    for (int v = 0; v < dim; ++v) {
        alpha_sum_f += shareds[0].alphas[v];
        shared_part[v] = orc_fast_lgamma(shareds[0].alphas[v]);
    }
    double alpha_sum_d = alpha_sum_f;
    shared_part[dim] = orc_fast_lgamma(alpha_sum_f);
    for (int k = 0; k < f->K; ++k) {
        if (!f->i0[k]) continue;
        for (int v = 0; v < dim; ++v)
            scores[v] += orc_fast_lgamma(
                shareds[0].alphas[v] + (float)f->cnt[(size_t)k * dim + v])
                       - shared_part[v];
        scores[dim] += shared_part[dim]
                     - orc_fast_lgamma(alpha_sum_f + (float)f->i0[k]);
    }
    scores_out[0] = orc_vector_sum((size_t)dim + 1, scores);
    for (size_t i = 1; i < n; ++i) {
        for (int v = 0; v < dim; ++v) {
            const float old_alpha = shareds[i - 1].alphas[v];
            const float new_alpha = shareds[i].alphas[v];
            if (new_alpha == old_alpha) continue;
            /* _update(value, old_alpha, new_alpha) */
            shared_part[v] = orc_fast_lgamma(new_alpha);
            alpha_sum_d += (double)new_alpha - (double)old_alpha;
            const float alpha_sum = (float)alpha_sum_d;
            shared_part[dim] = orc_fast_lgamma(alpha_sum);
            scores[v] = 0;
            scores[dim] = 0;
            for (int k = 0; k < f->K; ++k) {   /* empty groups included */
                scores[v] += orc_fast_lgamma(
                    new_alpha + (float)f->cnt[(size_t)k * dim + v])
                           - shared_part[v];
                scores[dim] += shared_part[dim]
                             - orc_fast_lgamma(alpha_sum + (float)f->i0[k]);
            }
        }
        scores_out[i] = orc_vector_sum((size_t)dim + 1, scores);
    }
    free(scores); free(shared_part);
    orc_ftz_restore(saved);
}

/* PitmanYor::score_counts (src/clustering.cc:144-183) */
float orc_py_score_counts(float alpha, float d, const int * counts, size_t n) {
    unsigned saved = orc_ftz_enable();
    double score = 0.0;
    size_t sample_size = 0, nonempty_group_count = 0;
    for (size_t i = 0; i < n; ++i) {
        size_t count = (size_t)counts[i];
        if (!count) continue;
        if (count == 1) {
            score += orc_fast_log((alpha + d * nonempty_group_count)
                                  / (alpha + sample_size));
        } else if (count == 2) {
            score += orc_fast_log(
                ((alpha + d * nonempty_group_count) * (1 - d))
                / ((alpha + sample_size) * (alpha + sample_size + 1)));
        } else {
            score += orc_fast_log(alpha + d * nonempty_group_count);
            score += orc_fast_lgamma((1 - d) + (count - 1))
                   - orc_fast_lgamma(1 - d);
            score -= orc_fast_lgamma((alpha + sample_size) + count)
                   - orc_fast_lgamma(alpha + sample_size);
        }
        nonempty_group_count += 1;
        sample_size += count;
    }
    orc_ftz_restore(saved);
    return (float)score;
}

/* ------------------------------------------------------------------------ */
/* whole-path drivers                                                       */

void orc_mix_init_from_assignments(orc_mix * m, size_t n_rows,
                                   const uint32_t * const * values,
                                   const uint32_t * assign_packed,
                                   int nonempty_groups, int empty_groups,
                                   uint32_t * assign_global_out) {
    unsigned saved = orc_ftz_enable();
    int K = nonempty_groups + empty_groups;
    int * counts = calloc(K, sizeof(int));
    for (int fi = 0; fi < m->F; ++fi) {
        m->f[fi].K = 0;
        for (int k = 0; k < K; ++k) orc_mix_slave_append_empty(m, fi);
    }
    for (size_t i = 0; i < n_rows; ++i) {
        uint32_t g = assign_packed[i];
        counts[g] += 1;
        for (int fi = 0; fi < m->F; ++fi) group_add(&m->f[fi], g, values[fi][i]);
    }
    orc_mix_driver_init(m, counts, K);
    for (int fi = 0; fi < m->F; ++fi) cache_update_all(&m->f[fi]);
    orc_mix_tracker_init(m, K);
    if (assign_global_out)
        for (size_t i = 0; i < n_rows; ++i)
            assign_global_out[i] = m->p2g[assign_packed[i]];
    free(counts);
    orc_ftz_restore(saved);
}

/* benchmarks/mixture.cc:104-115, the loop the reference's "cells/us" figure
 * times: remove the value from its group, score_value accumulating into a
 * vector that is zeroed every eight iterations, add it back.  Feature 0 of
 * `m`; values[i] sits in group groups[i]; returns a checksum of the scores
 * (so that the loop cannot be optimised away). */
float orc_mixture_benchmark_loop(orc_mix * m, size_t n_values,
                                 const uint32_t * values,
                                 const uint32_t * groups, size_t iters) {
    unsigned saved = orc_ftz_enable();
    feat * f = &m->f[0];
    float * scores = malloc(sizeof(float) * (f->K + 1));
    float check = 0.f;
    for (size_t i = 0; i < iters / 8; ++i) {
        for (int k = 0; k < f->K; ++k) scores[k] = 0.f;
        for (size_t j = 0; j < 8; ++j) {
            const size_t at = (8 * i + j) % n_values;
            orc_mix_slave_remove_value(m, 0, (int)groups[at], values[at]);
            feat_score_value(f, values[at], scores);
            orc_mix_slave_add_value(m, 0, (int)groups[at], values[at]);
        }
        check += scores[0];
    }
    free(scores);
    orc_ftz_restore(saved);
    return check;
}

/* Adopt a state that was produced elsewhere (tests: the engine's state after
 * some sweeps, so that the oracle can follow it from there): K groups in the
 * given slot order -- sizes, per-feature statistics as
 * orc_mix_slave_get_group lays them out (one block of K * words per feature),
 * the id maps of the tracker -- then the caches as init() builds them
 * (clustering.hpp:151-161, mixture.hpp:354-359). */
void orc_mix_load_state(orc_mix * m, int K, const int32_t * counts,
                        const uint32_t * const * group_words,
                        const uint32_t * p2g, uint32_t global_size) {
    unsigned saved = orc_ftz_enable();
    for (int fi = 0; fi < m->F; ++fi) {
        feat * f = &m->f[fi];
        f->K = 0;
        for (int k = 0; k < K; ++k) orc_mix_slave_append_empty(m, fi);
        const int kind = f->sh.kind;
        const size_t words = is_cat(kind) ? 1 + (size_t)f->sh.dim
                           : (kind == ORC_BB || kind == ORC_BNB) ? 2 : 3;
        for (int k = 0; k < K; ++k) {
            const uint32_t * w = group_words[fi] + (size_t)k * words;
            f->i0[k] = (int32_t)w[0];
            if (is_cat(kind)) {
                memcpy(f->cnt + (size_t)k * f->sh.dim, w + 1,
                       4 * (size_t)f->sh.dim);
            } else if (kind == ORC_BB || kind == ORC_BNB) {
                f->i1[k] = (int32_t)w[1];
            } else if (kind == ORC_GP) {
                f->i1[k] = (int32_t)w[1];
                f->f0[k] = u2f(w[2]);
            } else {
                f->f0[k] = u2f(w[1]);
                f->f1[k] = u2f(w[2]);
            }
        }
    }
    orc_mix_driver_init(m, counts, K);
    for (int fi = 0; fi < m->F; ++fi) cache_update_all(&m->f[fi]);
    /* the tracker: ids handed out so far, the live ones at their slots */
    orc_mix_tracker_init(m, 0);
    if ((int)global_size + 1 > m->g2p_cap) {
        m->g2p_cap = (int)global_size + 16;
        m->g2p = realloc(m->g2p, sizeof(int32_t) * m->g2p_cap);
    }
    if (K + 1 > m->p2g_cap) {
        m->p2g_cap = K + 16;
        m->p2g = realloc(m->p2g, sizeof(uint32_t) * m->p2g_cap);
    }
    for (uint32_t g = 0; g < global_size; ++g) m->g2p[g] = -1;
    for (int k = 0; k < K; ++k) {
        m->p2g[k] = p2g[k];
        m->g2p[p2g[k]] = k;
    }
    m->p2g_size = K;
    m->global_size = global_size;
    orc_ftz_restore(saved);
}

void orc_mix_gibbs_sequential(orc_mix * m, size_t row_begin, size_t row_end,
                              const uint32_t * const * values,
                              uint32_t * assign, uint32_t * rng_state) {
    unsigned saved = orc_ftz_enable();
    int scap = m->K + 64;
    float * scores = malloc(sizeof(float) * scap);
    for (size_t i = row_begin; i < row_end; ++i) {
        int g = (int)orc_mix_global_to_packed(m, assign[i]);
        int removed = orc_mix_driver_remove_value(m, g);
        for (int fi = 0; fi < m->F; ++fi)
            orc_mix_slave_remove_value(m, fi, g, values[fi][i]);
        if (removed) {
            for (int fi = 0; fi < m->F; ++fi) feat_remove_group(&m->f[fi], g);
            orc_mix_tracker_remove_group(m, (uint32_t)g);
        }
        if (m->K > scap) {
            scap = 2 * m->K;
            scores = realloc(scores, sizeof(float) * scap);
        }
        orc_mix_driver_score_value(m, scores);
        for (int fi = 0; fi < m->F; ++fi)
            feat_score_value(&m->f[fi], values[fi][i], scores);
        int g2 = (int)orc_sample_from_scores_overwrite(rng_state, m->K, scores);
        int added = orc_mix_driver_add_value(m, g2);
        for (int fi = 0; fi < m->F; ++fi)
            orc_mix_slave_add_value(m, fi, g2, values[fi][i]);
        if (added) {
            for (int fi = 0; fi < m->F; ++fi) feat_add_group(&m->f[fi]);
            orc_mix_tracker_add_group(m);
        }
        assign[i] = orc_mix_packed_to_global(m, (uint32_t)g2);
    }
    free(scores);
    orc_ftz_restore(saved);
}

/* The initialisation loop of examples/mixture/main.py: rows are ADDED one at a
 * time, nothing is removed.  prior_only = 0: compress_seq_gibbs (main.py:
 * 265-270), every row is scored against the groups built so far
 * (mixture.score_value) and sampled; prior_only = 1: compress_gibbs
 * (main.py:227-232), the clustering model's score alone
 * (mixture.clustering.score_value).  One engine step per row. */
void orc_mix_init_sequential(orc_mix * m, size_t row_begin, size_t row_end,
                             const uint32_t * const * values,
                             uint32_t * assign, uint32_t * rng_state,
                             int prior_only) {
    unsigned saved = orc_ftz_enable();
    int scap = m->K + 64;
    float * scores = malloc(sizeof(float) * scap);
    for (size_t i = row_begin; i < row_end; ++i) {
        if (m->K > scap) {
            scap = 2 * m->K;
            scores = realloc(scores, sizeof(float) * scap);
        }
        orc_mix_driver_score_value(m, scores);
        for (int fi = 0; !prior_only && fi < m->F; ++fi)
            feat_score_value(&m->f[fi], values[fi][i], scores);
        int g2 = (int)orc_sample_from_scores_overwrite(rng_state, m->K, scores);
        int added = orc_mix_driver_add_value(m, g2);
        for (int fi = 0; fi < m->F; ++fi)
            orc_mix_slave_add_value(m, fi, g2, values[fi][i]);
        if (added) {
            for (int fi = 0; fi < m->F; ++fi) feat_add_group(&m->f[fi]);
            orc_mix_tracker_add_group(m);
        }
        assign[i] = orc_mix_packed_to_global(m, (uint32_t)g2);
    }
    free(scores);
    orc_ftz_restore(saved);
}

/* Batch-semantics scores of one row whose current group is `g`: the state at
 * entry with the row itself taken out.  Nothing is mutated.
 *   n_g >= 2: group order unchanged, entry g re-derived from (stats - row)
 *             exactly as remove_value + cache refresh would leave it.
 *   n_g == 1: the group disappears as MixtureDriver::remove_value does it
 *             (last group moves into slot g, one group fewer) and the empty
 *             groups' prior is re-derived with one non-empty group fewer. */
int orc_mix_batch_row_scores(const orc_mix * m, const uint32_t * x, uint32_t g,
                             float * scores) {
    const int K = m->K;
    const int singleton = (m->counts[g] == 1);
    const int Kl = singleton ? K - 1 : K;
    const float shift =
        -orc_fast_log((float)(uint64_t)(m->sample_size - 1) + m->alpha);
    float empty_score = 0.f;
    if (singleton)
        empty_score = py_empty_score(m->alpha, m->d,
                                     K - m->n_empty - 1, m->n_empty);
    for (int k = 0; m->cluster && k < Kl; ++k) {
        /* generic driver after remove_value: sizes as they stand, one row
         * and (for a singleton) one non-empty group fewer */
        int src = (singleton && k == (int)g) ? K - 1 : k;
        int n = m->counts[src] - ((!singleton && k == (int)g) ? 1 : 0);
        scores[k] = orc_le_score_add_value(
            m->dataset_size, n, K - m->n_empty - singleton,
            (int)m->sample_size - 1, m->n_empty);
    }
    for (int k = 0; !m->cluster && k < Kl; ++k) {
        int src = (singleton && k == (int)g) ? K - 1 : k;
        float c;
        if (!singleton && k == (int)g) {
            c = orc_fast_log((float)(m->counts[g] - 1) - m->d);
        } else if (singleton && m->counts[src] == 0) {
            c = empty_score;
        } else {
            c = m->shifted[src];
        }
        scores[k] = c + shift;
    }
    for (int fi = 0; fi < m->F; ++fi) {
        const feat * f = &m->f[fi];
        uint32_t v = x[fi];
        float lf = f->sh.kind == ORC_GP ? orc_fast_log_factorial(v) : 0.f;
        /* the self-removed entry for slot g */
        float S_g = 0, H_g = 0;
        scorer4 s_g = {0, 0, 0, 0};
        if (!singleton) {
            if (is_cat(f->sh.kind)) {
                H_g = orc_fast_log(f->alpha_sum + (float)(f->i0[g] - 1));
                if (f->sh.kind == ORC_DPD && v == 0xFFFFFFFFu)
                    S_g = cat_entry(f, v, g);
                else
                    S_g = orc_fast_log(cat_prior(f, v) + (float)(
                        f->cnt[(size_t)g * f->sh.dim + v] - 1));
            } else {
                feat tmp = *f; /* one-group scratch copy of the stats */
                int32_t i0 = f->i0[g], i1 = f->i1[g];
                float f0 = f->f0[g], f1 = f->f1[g];
                tmp.i0 = &i0; tmp.i1 = &i1; tmp.f0 = &f0; tmp.f1 = &f1;
                group_remove(&tmp, 0, v);
                s_g = scorer_init(&f->sh, i0, i1, f0, f1);
            }
        }
        for (int k = 0; k < Kl; ++k) {
            int src = (singleton && k == (int)g) ? K - 1 : k;
            int patched = (!singleton && k == (int)g);
            if (is_cat(f->sh.kind)) {
                float S = patched ? S_g : cat_entry(f, v, src);
                float H = patched ? H_g : f->c0[src];
                scores[k] = (scores[k] + S) - H;
            } else {
                scorer4 s = {f->c0[src], f->c1[src], f->c2[src], f->c3[src]};
                if (patched) s = s_g;
                scores[k] += noncat_term(f->sh.kind, s, v, lf);
            }
        }
    }
    return Kl;
}

/* ---- the batch in phases (what a multi-GPU driver interleaves with its
 * all-reduce); orc_mix_gibbs_batch below is their composition -------------- */

/* phase 1: score + sample every row against the frozen state.  Row i uses
 * engine draw (draw_base + row_offset + i). */
void orc_mix_batch_sample(const orc_mix * m, size_t row_begin, size_t row_end,
                          const uint32_t * const * values,
                          const uint32_t * assign, uint32_t seed_state,
                          uint64_t draw_base, uint64_t row_offset,
                          uint32_t * old_p, uint32_t * new_p) {
    unsigned saved = orc_ftz_enable();
    const int K = m->K;
    float * scores = malloc(sizeof(float) * (K + 1));
    uint32_t x[64];
    for (size_t i = row_begin; i < row_end; ++i) {
        uint32_t g = orc_mix_global_to_packed(m, assign[i]);
        for (int fi = 0; fi < m->F; ++fi) x[fi] = values[fi][i];
        int Kl = orc_mix_batch_row_scores(m, x, g, scores);
        uint32_t st = orc_rng_jump(seed_state, draw_base + row_offset + i);
        float u = orc_sample_unif01(&st);
        uint32_t g2 = (uint32_t)orc_sample_from_scores_u(Kl, scores, u);
        if (Kl != K && g2 == g) g2 = (uint32_t)(K - 1); /* slot g held K-1 */
        old_p[i - row_begin] = g;
        new_p[i - row_begin] = g2;
    }
    free(scores);
    orc_ftz_restore(saved);
}

/* The statistics split in two parts for the multi-rank exchange: integers
 * that sum over ranks (group sizes, DD/DPD/BB counts, GP count and sum) and
 * the order-dependent ones, which are replayed in global row order (all of
 * NICH's count/mean/count_times_variance, nich.hpp:125-165; GP's log_prod,
 * gp.hpp:115,134). */
static void group_apply_part(feat * f, int k, uint32_t value, int add,
                             int part) {
    const int kind = f->sh.kind;
    if (kind == ORC_NICH) {
        if (!(part & ORC_PART_ORDERED)) return;
        if (add) group_add(f, k, value); else group_remove(f, k, value);
        return;
    }
    if (kind == ORC_GP) {
        const int32_t i0 = f->i0[k], i1 = f->i1[k];
        const float f0 = f->f0[k];
        if (add) group_add(f, k, value); else group_remove(f, k, value);
        if (!(part & ORC_PART_SUMMED)) { f->i0[k] = i0; f->i1[k] = i1; }
        if (!(part & ORC_PART_ORDERED)) f->f0[k] = f0;
        return;
    }
    if (!(part & ORC_PART_SUMMED)) return;
    if (add) group_add(f, k, value); else group_remove(f, k, value);
}

/* phase 2: apply the moves in row order (remove, then add, per row) */
void orc_mix_apply_moves_part(orc_mix * m, size_t row_begin, size_t row_end,
                              const uint32_t * const * values,
                              uint32_t * assign, const uint32_t * old_p,
                              const uint32_t * new_p, int part) {
    unsigned saved = orc_ftz_enable();
    for (size_t i = row_begin; i < row_end; ++i) {
        uint32_t g = old_p[i - row_begin], g2 = new_p[i - row_begin];
        if (part & ORC_PART_SUMMED) {
            m->counts[g] -= 1;
            m->counts[g2] += 1;
            assign[i] = m->p2g[g2];
        }
        for (int fi = 0; fi < m->F; ++fi) {
            group_apply_part(&m->f[fi], g, values[fi][i], 0, part);
            group_apply_part(&m->f[fi], g2, values[fi][i], 1, part);
        }
    }
    orc_ftz_restore(saved);
}
void orc_mix_apply_moves(orc_mix * m, size_t row_begin, size_t row_end,
                         const uint32_t * const * values, uint32_t * assign,
                         const uint32_t * old_p, const uint32_t * new_p) {
    orc_mix_apply_moves_part(m, row_begin, row_end, values, assign, old_p,
                             new_p, ORC_PART_SUMMED | ORC_PART_ORDERED);
}

/* the order-dependent statistics over an event list gathered from all ranks
 * (rank order == global row order): row i leaves slot old_p[i] (old_p == NULL:
 * rows are only added) and joins new_p[i]; 0xFFFFFFFF marks padding.
 * values[fi][i] is row i's value of feature fi (only read for NICH/GP
 * features).  reset != 0 zeroes those statistics first (initial load). */
void orc_mix_replay_ordered(orc_mix * m, size_t n,
                            const uint32_t * const * values,
                            const uint32_t * old_p, const uint32_t * new_p,
                            int reset) {
    unsigned saved = orc_ftz_enable();
    for (int fi = 0; fi < m->F; ++fi) {
        feat * f = &m->f[fi];
        const int kind = f->sh.kind;
        if (kind != ORC_NICH && kind != ORC_GP) continue;
        if (reset) {
            for (int k = 0; k < m->K; ++k) {
                f->f0[k] = 0.f;
                if (kind == ORC_NICH) { f->i0[k] = 0; f->f1[k] = 0.f; }
            }
        }
        for (size_t i = 0; i < n; ++i) {
            if (new_p[i] == 0xFFFFFFFFu) continue;
            if (old_p)
                group_apply_part(f, (int)old_p[i], values[fi][i], 0,
                                 ORC_PART_ORDERED);
            group_apply_part(f, (int)new_p[i], values[fi][i], 1,
                             ORC_PART_ORDERED);
        }
    }
    orc_ftz_restore(saved);
}

/* integer statistics as words: counts[K] | per feature i0[K] i1[K]
 * (categorical: cnt[K][dim]) -- the layout of dist_gibbs_stat_words() */
size_t orc_mix_stat_words(const orc_mix * m) {
    size_t n = (size_t)m->K;
    for (int fi = 0; fi < m->F; ++fi)
        n += 2 * (size_t)m->K
           + (is_cat(m->f[fi].sh.kind) ? (size_t)m->K * m->f[fi].sh.dim : 0);
    return n;
}
static void stat_copy(orc_mix * m, int32_t * w, int to_words) {
    size_t K = (size_t)m->K;
#define ORC_CP(live, n) do {                          \
        if (to_words) { memcpy(w, live, 4 * (n)); }       \
        else { memcpy(live, w, 4 * (n)); }                \
        w += (n);                                         \
    } while (0)
    ORC_CP(m->counts, K);
    for (int fi = 0; fi < m->F; ++fi) {
        feat * f = &m->f[fi];
        ORC_CP(f->i0, K);
        ORC_CP(f->i1, K);
        if (is_cat(f->sh.kind)) ORC_CP(f->cnt, K * (size_t)f->sh.dim);
    }
#undef ORC_CP
}
void orc_mix_export_stats(const orc_mix * m, int32_t * words) {
    stat_copy((orc_mix *)m, words, 1);
}
void orc_mix_import_stats(orc_mix * m, const int32_t * words) {
    stat_copy(m, (int32_t *)words, 0);
}

/* phases 3 + 4: normalise the group set against the counts at batch entry
 * and rebuild the caches.  Groups that lost their last member are
 * swap-removed in descending slot order; one new empty group is appended for
 * every previously empty group that gained members. */
void orc_mix_batch_finish(orc_mix * m, const int32_t * snap_counts) {
    unsigned saved = orc_ftz_enable();
    const int K = m->K;
    int created = 0;
    for (int k = 0; k < K; ++k)
        if (snap_counts[k] == 0 && m->counts[k] > 0) created += 1;
    for (int k = K - 1; k >= 0; --k) {
        if (snap_counts[k] > 0 && m->counts[k] == 0) {
            int last = m->K - 1;
            if (k != last) m->counts[k] = m->counts[last];
            m->K = last;
            for (int fi = 0; fi < m->F; ++fi) feat_remove_group(&m->f[fi], k);
            orc_mix_tracker_remove_group(m, (uint32_t)k);
        }
    }
    for (int c = 0; c < created; ++c) {
        py_reserve(m, m->K + 1);
        m->counts[m->K] = 0;
        m->K += 1;
        for (int fi = 0; fi < m->F; ++fi) {
            feat * f = &m->f[fi];
            feat_reserve(f, f->K + 1);
            group_init(f, f->K);
            f->K += 1;
        }
        orc_mix_tracker_add_group(m);
    }
    /* caches are pure functions of the statistics */
    int * counts = malloc(sizeof(int) * (m->K + 1));
    memcpy(counts, m->counts, sizeof(int) * m->K);
    orc_mix_driver_init(m, counts, m->K);
    free(counts);
    for (int fi = 0; fi < m->F; ++fi) cache_update_all(&m->f[fi]);
    orc_ftz_restore(saved);
}

void orc_mix_gibbs_batch(orc_mix * m, size_t row_begin, size_t row_end,
                         const uint32_t * const * values, uint32_t * assign,
                         uint32_t seed_state, uint64_t draw_base) {
    const size_t B = row_end - row_begin;
    uint32_t * old_p = malloc(sizeof(uint32_t) * (B + 1));
    uint32_t * new_p = malloc(sizeof(uint32_t) * (B + 1));
    int32_t * snap_counts = malloc(sizeof(int32_t) * (m->K + 1));
    memcpy(snap_counts, m->counts, sizeof(int32_t) * m->K);
    orc_mix_batch_sample(m, row_begin, row_end, values, assign, seed_state,
                         draw_base, 0, old_p, new_p);
    orc_mix_apply_moves(m, row_begin, row_end, values, assign, old_p, new_p);
    orc_mix_batch_finish(m, snap_counts);
    free(snap_counts); free(old_p); free(new_p);
}
