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
    float p[4];       // hyper-parameters (dist_shared_t::p)
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
        e.c0 = -fast_lgamma(post_alpha)
             + post_alpha * (fast_log(post_inv_beta) + score_coeff);
        e.c1 = post_alpha;
        e.c2 = score_coeff;
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
DIST_HD float accumulate(int kind, float acc, const Entry & e, uint32_t value,
                         float lf) {
    switch (kind) {
    case DIST_DD:
    case DIST_DPD:
        return (acc + e.c1) - e.c0;
    case DIST_BB:
        return acc + (value ? e.c0 : e.c1);
    case DIST_GP: {
        const float fv = (float)value;
        return acc + (e.c0 + fast_lgamma(e.c1 + fv) - lf + e.c2 * fv);
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
DIST_HD float score_group(int kind, const Entry & e, uint32_t value, float lf) {
    switch (kind) {
    case DIST_DD:
    case DIST_DPD:
        return e.c1 - e.c0;
    case DIST_BB:
        return value ? e.c0 : e.c1;
    case DIST_GP: {
        const float fv = (float)value;
        return e.c0 + fast_lgamma(e.c1 + fv) - lf + e.c2 * fv;
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
DIST_HD Entry entry_after_remove(const SlaveView & s, int g, uint32_t value) {
    Entry e = {0.f, 0.f, 0.f, 0.f};
    if (is_cat(s.kind)) {
        e.c0 = fast_log(s.alpha_sum + (float)(s.i0[g] - 1));
        if (s.kind == DIST_DPD && value == DIST_DPD_OTHER) {
            e.c1 = s.other;
        } else {
            e.c1 = fast_log(
                s.prior[value]
                + (float)(s.cnt[(size_t)g * s.dim + value] - 1));
        }
        return e;
    }
    Stats st = load_stats(s, g);
    stats_remove(s.kind, st, value);
    return scorer_init(s.kind, s.p, st);
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
