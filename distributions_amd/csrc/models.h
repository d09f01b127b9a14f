// Component models of the hot path as device/host inline code over a
// structure-of-arrays view of one feature's groups.
//
// One feature ("slave") keeps, per group k, its sufficient statistics and the
// MixtureValueScorer cache of the reference, all as arrays over k in HBM:
//
//   kind   statistics                      cache (pure function of the stats)
//   DD     i0=count_sum, cnt[k][dim]       c0=log(A+n_k)   S[v][k]=log(a_v+n_kv)
//   DPD    i0=total,     cnt[k][dim]       c0=log(a+n_k)   S[v][k]=log(a*b_v+n_kv)
//   BB     i0=heads, i1=tails              c0=heads score  c1=tails score
//   GP     i0=count, i1=sum, f0=log_prod   c0=score c1=post_alpha c2=score_coeff
//   NICH   i0=count, f0=mean, f1=ctv       c0=score c1=log_coeff c2=precision c3=mean
//
// Reference: models/dd.hpp:346-472, dpd.hpp:376-578, bb.hpp:231-325,
// gp.hpp:243-334 + src/models/gp.cc:32-67, nich.hpp:290-385 +
// src/models/nich.cc:33-66 (paths relative to /root/reference).
#pragma once

#include "special.h"
#include "../../include/distributions_hip.h"

namespace dist {

struct SlaveView {
    int kind;
    int dim;          // categorical kinds: number of values
    float p[16];      // [0..3] hyper-parameters (dist_shared_t::p);
                      // GP: [4..9] the arguments below 2.5 that fast_lgamma
                      // can see ((alpha + s) + x, s + x <= 2) and [10..15]
                      // glibc's lgammaf of each (see gp_lgamma)
    float alpha_sum;  // DD: sum of alphas (dd.hpp:403-406); DPD: alpha
    float other;      // DPD: fast_log(alpha * beta0), score of OTHER
    int K;            // groups
    int cap;          // allocated groups (= row stride of S)
    int32_t * i0;
    int32_t * i1;
    float * f0;
    float * f1;
    int32_t * cnt;    // [cap][dim]
    float * c0;
    float * c1;
    float * c2;
    float * c3;
    float * S;        // [dim][cap]
    const float * prior;  // [dim]: DD alphas[v]; DPD alpha * betas[v]
};

DIST_HD bool is_cat(int kind) { return kind == DIST_DD || kind == DIST_DPD; }
DIST_HD bool has_float_stats(int kind) {
    return kind == DIST_GP || kind == DIST_NICH;
}

// one group's statistics in registers
struct Stats {
    int32_t i0, i1;
    float f0, f1;
};
// one group's cache entry in registers; categorical kinds use c0 = shift,
// c1 = the table entry of the row's value
struct Entry {
    float c0, c1, c2, c3;
};

// Group::add_value for the scalar statistics (bb.hpp:102-107, gp.hpp:109-116,
// nich.hpp:125-133; the categorical count matrix is handled by the caller)
DIST_HD void stats_add(int kind, Stats & s, uint32_t value) {
    switch (kind) {
    case DIST_DD:
    case DIST_DPD:
        s.i0 += 1;
        break;
    case DIST_BB:
        if (value) s.i0 += 1; else s.i1 += 1;
        break;
    case DIST_GP:
        s.i0 = (int32_t)((uint32_t)s.i0 + 1u);
        s.i1 = (int32_t)((uint32_t)s.i1 + value);
        s.f0 += fast_log_factorial(value);
        break;
    case DIST_BNB:   // bnb.hpp:106-112
        s.i0 = (int32_t)((uint32_t)s.i0 + 1u);
        s.i1 = (int32_t)((uint32_t)s.i1 + value);
        break;
    default: {  // DIST_NICH
        const float x = u2f(value);
        s.i0 += 1;
        const float delta = x - s.f0;
        s.f0 += delta / (float)s.i0;
        s.f1 += delta * (x - s.f0);
        break;
    }
    }
}

// Group::remove_value (bb.hpp:117-122, gp.hpp:128-135, nich.hpp:146-165)
DIST_HD void stats_remove(int kind, Stats & s, uint32_t value) {
    switch (kind) {
    case DIST_DD:
    case DIST_DPD:
        s.i0 -= 1;
        break;
    case DIST_BB:
        if (value) s.i0 -= 1; else s.i1 -= 1;
        break;
    case DIST_GP:
        s.i0 = (int32_t)((uint32_t)s.i0 - 1u);
        s.i1 = (int32_t)((uint32_t)s.i1 - value);
        s.f0 -= fast_log_factorial(value);
        break;
    case DIST_BNB:   // bnb.hpp:124-130
        s.i0 = (int32_t)((uint32_t)s.i0 - 1u);
        s.i1 = (int32_t)((uint32_t)s.i1 - value);
        break;
    default: {  // DIST_NICH
        const float x = u2f(value);
        const float total = s.f0 * (float)s.i0;
        const float delta = x - s.f0;
        s.i0 -= 1;
        if (s.i0 == 0) {
            s.f0 = 0.f;
        } else {
            s.f0 = (total - x) / (float)s.i0;
        }
        if (s.i0 <= 1) {
            s.f1 = 0.f;
        } else {
            s.f1 -= delta * (x - s.f0);
        }
        break;
    }
    }
}

// fast_lgamma as GammaPoisson calls it.  Below 2.5 the reference calls
// glibc's lgammaf (special.hpp:121-123).  With alpha fixed, only a handful of
// arguments below 2.5 exist -- (alpha + s) + x with s + x <= 2 -- so the host
// evaluates glibc's lgammaf on exactly those floats once and the device looks
// the result up: bit-identical to the reference for every alpha.
DIST_HD float gp_lgamma(float y, const float * p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (y < 2.5f) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (y == p[4 + i]) return p[10 + i];
    }
#else
    (void)p;
#endif
    return fast_lgamma(y);
}

// Model::Scorer::init over Shared::plus_group for the non-categorical kinds
// (bb.hpp:189-197; gp.hpp:56-61,198-207; nich.hpp:58-69,239-250)
DIST_HD Entry scorer_init(int kind, const float * p, const Stats & s) {
    Entry e = {0.f, 0.f, 0.f, 0.f};
    switch (kind) {
    case DIST_BB: {
        const float alpha = p[0] + (float)s.i0;
        const float beta = p[1] + (float)s.i1;
        e.c0 = fast_log(alpha / (alpha + beta));
        e.c1 = fast_log(beta / (alpha + beta));
        break;
    }
    case DIST_GP: {
        const float post_alpha = p[0] + (float)(uint32_t)s.i1;
        const float post_inv_beta = p[1] + (float)(uint32_t)s.i0;
        const float score_coeff = -fast_log(1.f + post_inv_beta);
        e.c0 = -gp_lgamma(post_alpha, p)
             + post_alpha * (fast_log(post_inv_beta) + score_coeff);
        e.c1 = post_alpha;
        e.c2 = score_coeff;
        break;
    }
    case DIST_BNB: {   // bnb.hpp:55-61 (plus_group), 200-215 (Scorer::init)
        const float r = p[2];
        const float post_alpha = p[0] + r * (float)(uint32_t)s.i0;
        const float post_beta = p[1] + (float)(uint32_t)s.i1;
        const float alpha = post_alpha + r;
        e.c0 = fast_lgamma(post_alpha + post_beta) - fast_lgamma(post_alpha)
             - fast_lgamma(post_beta) + fast_lgamma(alpha);
        e.c1 = post_beta;
        e.c2 = alpha;
        break;
    }
    case DIST_NICH: {
        const float mu = p[0], kappa = p[1], sigmasq = p[2], nu = p[3];
        const float count = (float)s.i0, mean = s.f0, ctv = s.f1;
        const float mu_1 = mu - mean;
        const float post_kappa = kappa + count;
        const float post_mu = (kappa * mu + mean * count) / post_kappa;
        const float post_nu = nu + count;
        const float post_sigmasq = 1.f / post_nu * (
            nu * sigmasq + ctv + (count * kappa * mu_1 * mu_1) / post_kappa);
        const float lambda = post_kappa / ((post_kappa + 1.f) * post_sigmasq);
        e.c0 = fast_lgamma_nu(post_nu)
             + 0.5f * fast_log(lambda / (3.14159265358979f * post_nu));
        e.c1 = -0.5f * post_nu - 0.5f;
        e.c2 = lambda / post_nu;
        e.c3 = post_mu;
        break;
    }
    default:
        break;
    }
    return e;
}

// acc (+)= log p(value | group with cache entry e), in the reference's order:
//   DD/DPD  (acc + S) - shift      dd.hpp:433-445 -> vector_math.cc:160-168
//                                  (release build: add first, then subtract)
//   BB      acc + (v ? heads : tails)                      bb.hpp:303-313
//   GP      acc + (((score + lgamma(a+v)) - logfact(v)) + coeff*v)   gp.cc:57-66
//   NICH    acc + (score + log_coeff*log(1 + prec*(v-mean)^2))     nich.cc:60-66
// `lf` = fast_log_factorial(value) for GP (hoisted like gp.cc:56).
// `p` = the feature's parameter block (SlaveView::p; GP reads its lgamma
// table from it, see gp_lgamma).
DIST_HD float accumulate(int kind, float acc, const Entry & e, uint32_t value,
                         float lf, const float * p) {
    switch (kind) {
    case DIST_DD:
    case DIST_DPD:
        return (acc + e.c1) - e.c0;
    case DIST_BB:
        return acc + (value ? e.c0 : e.c1);
    case DIST_GP: {
        const float fv = (float)value;
        return acc + (e.c0 + gp_lgamma(e.c1 + fv, p) - lf + e.c2 * fv);
    }
    case DIST_BNB: {   // bnb.hpp:316-327
        const float beta = e.c1 + (float)value;
        return acc + (e.c0 + fast_lgamma(beta) - fast_lgamma(beta + e.c2));
    }
    default: {  // DIST_NICH
        const float x = u2f(value);
        const float d = x - e.c3;
        const float temp = 1.f + e.c2 * (d * d);
        return acc + (e.c0 + e.c1 * fast_log(temp));
    }
    }
}

// score_value_group (dd.hpp:423-431, bb.hpp:293-301, gp.hpp:300-310,
// nich.hpp:351-360, dpd.hpp:499-515)
DIST_HD float score_group(int kind, const Entry & e, uint32_t value, float lf,
                          const float * p) {
    switch (kind) {
    case DIST_DD:
    case DIST_DPD:
        return e.c1 - e.c0;
    case DIST_BB:
        return value ? e.c0 : e.c1;
    case DIST_GP: {
        const float fv = (float)value;
        return e.c0 + gp_lgamma(e.c1 + fv, p) - lf + e.c2 * fv;
    }
    case DIST_BNB: {   // bnb.hpp:304-314
        const float beta = e.c1 + (float)value;
        return e.c0 + fast_lgamma(beta) - fast_lgamma(beta + e.c2);
    }
    default: {
        const float x = u2f(value);
        const float d = x - e.c3;
        const float temp = 1.f + e.c2 * (d * d);
        return e.c0 + e.c1 * fast_log(temp);
    }
    }
}

DIST_HD Stats load_stats(const SlaveView & s, int k) {
    Stats st;
    st.i0 = s.i0[k];
    st.i1 = s.i1[k];
    st.f0 = s.f0[k];
    st.f1 = s.f1[k];
    return st;
}
DIST_HD void store_stats(const SlaveView & s, int k, const Stats & st) {
    s.i0[k] = st.i0;
    s.i1[k] = st.i1;
    s.f0[k] = st.f0;
    s.f1[k] = st.f1;
}

// the cached entry of group k for a row whose value is `value`
DIST_HD Entry load_entry(const SlaveView & s, int k, uint32_t value) {
    Entry e;
    e.c0 = s.c0[k];
    if (is_cat(s.kind)) {
        // dpd.hpp:534-542: OTHER scores with the scalar fast_log(alpha*beta0)
        e.c1 = (s.kind == DIST_DPD && value == DIST_DPD_OTHER)
                   ? s.other
                   : s.S[(size_t)value * s.cap + k];
        e.c2 = 0.f;
        e.c3 = 0.f;
    } else {
        e.c1 = s.c1[k];
        e.c2 = s.c2[k];
        e.c3 = s.c3[k];
    }
    return e;
}

// the entry group g would have after remove_value(value): what
// MixtureSlave::remove_value leaves in the cache (mixture.hpp:386-398;
// dd.hpp:390-397,458-467; gp.hpp:275-282; nich.hpp:335-342; bb.hpp:267-274)
// (`kind`: the view's kind, for callers that know it at compile time and read
// the view in place)
DIST_HD Entry entry_after_remove(const SlaveView & s, int g, uint32_t value,
                                 int kind) {
    Entry e = {0.f, 0.f, 0.f, 0.f};
    if (is_cat(kind)) {
        e.c0 = fast_log(s.alpha_sum + (float)(s.i0[g] - 1));
        if (kind == DIST_DPD && value == DIST_DPD_OTHER) {
            e.c1 = s.other;
        } else {
            e.c1 = fast_log(
                s.prior[value]
                + (float)(s.cnt[(size_t)g * s.dim + value] - 1));
        }
        return e;
    }
    Stats st = load_stats(s, g);
    stats_remove(kind, st, value);
    return scorer_init(kind, s.p, st);
}
DIST_HD Entry entry_after_remove(const SlaveView & s, int g, uint32_t value) {
    return entry_after_remove(s, g, value, s.kind);
}

// MixtureValueScorer::update_group for one (group, value) cell of a
// categorical feature, or the whole entry of a scalar one
DIST_HD void refresh_cat_cell(const SlaveView & s, int k, int v) {
    s.S[(size_t)v * s.cap + k] =
        fast_log(s.prior[v] + (float)s.cnt[(size_t)k * s.dim + v]);
}
DIST_HD void refresh_shift(const SlaveView & s, int k) {
    s.c0[k] = fast_log(s.alpha_sum + (float)s.i0[k]);
}
DIST_HD void refresh_scalar_entry(const SlaveView & s, int k) {
    const Entry e = scorer_init(s.kind, s.p, load_stats(s, k));
    s.c0[k] = e.c0;
    s.c1[k] = e.c1;
    s.c2[k] = e.c2;
    s.c3[k] = e.c3;
}

// Shared::plus_group for NICH (nich.hpp:58-69), returning post kappa, mu, nu,
// sigmasq
DIST_HD void nich_plus_group(const float * p, const Stats & s, float & pk,
                             float & pmu, float & pnu, float & psig) {
    const float mu = p[0], kappa = p[1], sigmasq = p[2], nu = p[3];
    const float count = (float)s.i0, mean = s.f0, ctv = s.f1;
    const float mu_1 = mu - mean;
    pk = kappa + count;
    pmu = (kappa * mu + mean * count) / pk;
    pnu = nu + count;
    psig = 1.f / pnu * (nu * sigmasq + ctv + (count * kappa * mu_1 * mu_1) / pk);
}

// Group::score_data for the scalar kinds (bb.hpp:141-151, gp.hpp:155-164,
// nich.hpp:190-202): log marginal likelihood of the group's data
DIST_HD float scalar_group_score_data(int kind, const float * p,
                                      const Stats & s) {
    switch (kind) {
    case DIST_BB: {
        const float alpha = p[0] + (float)s.i0;
        const float beta = p[1] + (float)s.i1;
        float score = 0.f;
        score += fast_lgamma(alpha) - fast_lgamma(p[0]);
        score += fast_lgamma(beta) - fast_lgamma(p[1]);
        score += fast_lgamma(p[0] + p[1]) - fast_lgamma(alpha + beta);
        return score;
    }
    case DIST_BNB: {   // bnb.hpp:157-166
        const float pa = p[0] + p[2] * (float)(uint32_t)s.i0;
        const float pb = p[1] + (float)(uint32_t)s.i1;
        float score = fast_lgamma(p[0] + p[1]) - fast_lgamma(pa + pb);
        score += fast_lgamma(pa) - fast_lgamma(p[0]);
        score += fast_lgamma(pb) - fast_lgamma(p[1]);
        return score;
    }
    case DIST_GP: {
        const float post_alpha = p[0] + (float)(uint32_t)s.i1;
        const float post_inv_beta = p[1] + (float)(uint32_t)s.i0;
        float score = fast_lgamma(post_alpha) - fast_lgamma(p[0]);
        score += p[0] * fast_log(p[1]) - post_alpha * fast_log(post_inv_beta);
        score += -s.f0;
        return score;
    }
    default: {  // DIST_NICH
        float pk, pmu, pnu, psig;
        nich_plus_group(p, s, pk, pmu, pnu, psig);
        const float log_pi = 1.1447298858493991f;
        float score = fast_lgamma(0.5f * pnu) - fast_lgamma(0.5f * p[3]);
        score += 0.5f * fast_log(p[1] / pk);
        score += 0.5f * p[3] * (fast_log(p[3] * p[2]))
               - 0.5f * pnu * fast_log(pnu * psig);
        score += -0.5f * (float)s.i0 * log_pi;
        return score;
    }
    }
}

// One group's contribution to MixtureDataScorer::score_data for the scalar
// kinds (bb.hpp:207-229, gp.hpp:220-241, nich.hpp:262-288), as the separate
// float terms the reference adds to its accumulator; returns how many.
DIST_HD int scalar_mixture_score_terms(int kind, const float * p,
                                       const Stats & s, float (&t)[4]) {
    switch (kind) {
    case DIST_BB: {   // every group, empty ones included
        const float shared_part =
            + fast_lgamma(p[0] + p[1]) - fast_lgamma(p[0]) - fast_lgamma(p[1]);
        const float alpha = p[0] + (float)s.i0;
        const float beta = p[1] + (float)s.i1;
        const float group_part =
            + fast_lgamma(alpha) + fast_lgamma(beta) - fast_lgamma(alpha + beta);
        t[0] = shared_part + group_part;
        return 1;
    }
    case DIST_BNB: {   // bnb.hpp:226-245
        if (s.i0 == 0) return 0;
        const float shared_part =
            fast_lgamma(p[0] + p[1]) - fast_lgamma(p[0]) - fast_lgamma(p[1]);
        const float pa = p[0] + p[2] * (float)(uint32_t)s.i0;
        const float pb = p[1] + (float)(uint32_t)s.i1;
        t[0] = fast_lgamma(pa) + fast_lgamma(pb) - fast_lgamma(pa + pb);
        t[1] = shared_part;
        return 2;
    }
    case DIST_GP: {
        if (s.i0 == 0) return 0;
        const float alpha_part = fast_lgamma(p[0]);
        const float beta_part = p[0] * fast_log(p[1]);
        const float post_alpha = p[0] + (float)(uint32_t)s.i1;
        const float post_inv_beta = p[1] + (float)(uint32_t)s.i0;
        t[0] = fast_lgamma(post_alpha) - alpha_part;
        t[1] = beta_part - post_alpha * fast_log(post_inv_beta);
        t[2] = -s.f0;
        return 3;
    }
    default: {  // DIST_NICH
        if (s.i0 == 0) return 0;
        const float nu_part = fast_lgamma(0.5f * p[3]);
        const float kappa_part = 0.5f * fast_log(p[1]);
        const float sigmasq_part = 0.5f * p[3] * fast_log(p[3] * p[2]);
        const float log_pi = 1.1447298858493991f;
        float pk, pmu, pnu, psig;
        nich_plus_group(p, s, pk, pmu, pnu, psig);
        t[0] = fast_lgamma(0.5f * pnu) - nu_part;
        t[1] = kappa_part - 0.5f * fast_log(pk);
        t[2] = sigmasq_part - 0.5f * pnu * fast_log(pnu * psig);
        t[3] = -0.5f * log_pi * (float)s.i0;
        return 4;
    }
    }
}

// vector_sum (vector_math.cc:85-93) in the association of the reference's
// release build: four lane accumulators (element i in lane i mod 4) over the
// first 4*floor(n/4) elements, (lane1 + lane3) + (lane0 + lane2), then the
// tail in order; fewer than four elements in order
DIST_HD float vector_sum_as_built(size_t n, const float * x) {
    if (n < 4) {
        float s = 0.f;
        for (size_t i = 0; i < n; ++i) s += x[i];
        return s;
    }
    float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
    const size_t body = n & ~(size_t)3;
    for (size_t i = 0; i < body; i += 4) {
        l0 += x[i]; l1 += x[i + 1]; l2 += x[i + 2]; l3 += x[i + 3];
    }
    float s = (l1 + l3) + (l0 + l2);
    for (size_t i = body; i < n; ++i) s += x[i];
    return s;
}

// PitmanYor::score_counts (src/clustering.cc:152-183): the term of one
// non-empty group given how many non-empty groups and rows precede it
DIST_HD double py_score_counts_term(float alpha, float d, int count,
                                    unsigned long long nonempty_before,
                                    unsigned long long rows_before) {
    const float ne = (float)nonempty_before;
    const float ss = (float)rows_before;
    if (count == 1) {
        return fast_log((alpha + d * ne) / (alpha + ss));
    }
    if (count == 2) {
        return fast_log(((alpha + d * ne) * (1 - d))
                        / ((alpha + ss) * (alpha + ss + 1)));
    }
    double score = 0.0;
    score += fast_log(alpha + d * ne);
    score += fast_lgamma((1 - d) + (float)(unsigned long long)(count - 1))
           - fast_lgamma(1 - d);
    score -= fast_lgamma((alpha + ss) + (float)(unsigned long long)count)
           - fast_lgamma(alpha + ss);
    return score;
}

// Clustering<int>::PitmanYor cached scores (clustering.hpp:215-230)
DIST_HD float py_nonempty_score(int count, float d) {
    return fast_log((float)count - d);
}
DIST_HD float py_empty_score(float alpha, float d, int nonempty, int empty) {
    const float numer = alpha + d * (float)nonempty;
    const float denom = (float)empty;
    return fast_log(numer / denom);
}
// clustering.hpp:202: shift = -fast_log(sample_size + alpha)
DIST_HD float py_shift(long long sample_size, float alpha) {
    return -fast_log((float)(unsigned long long)sample_size + alpha);
}
// Clustering<int>::LowEntropy (clustering.hpp:245-331).  dataset_size is the
// model's only parameter; nonempty_group_count does not enter its scores.
// _approximate_postpred_correction, clustering.hpp:318-327
DIST_HD float le_postpred_correction(float sample_size, int dataset_size) {
    const float exponent =
        0.45f - 0.1f / sample_size - 0.1f / (float)dataset_size;
    const float scale = (float)dataset_size / sample_size;
    return fast_log(scale) * exponent;
}
// score_add_value, clustering.hpp:267-292
DIST_HD float le_score_add_value(int dataset_size, int group_size,
                                 int sample_size, int empty) {
    if (group_size == 0) {
        float score = -fast_log((float)empty);
        if (sample_size + 1 < dataset_size)
            score += le_postpred_correction((float)(sample_size + 1),
                                            dataset_size);
        return score;
    }
    const float bigger = 1.f + (float)group_size;
    if (group_size > 10000) return 1.f + fast_log(bigger);
    return fast_log(bigger / (float)group_size) * (float)group_size
         + fast_log(bigger);
}

// clustering.hpp:81-104
DIST_HD float py_score_add_value(float alpha, float d, int group_size,
                                 int nonempty, int sample_size, int empty) {
    if (group_size == 0) {
        const float numer = alpha + d * (float)nonempty;
        const float denom = ((float)sample_size + alpha) * (float)empty;
        return fast_log(numer / denom);
    }
    return fast_log(((float)group_size - d) / ((float)sample_size + alpha));
}

}  // namespace dist
