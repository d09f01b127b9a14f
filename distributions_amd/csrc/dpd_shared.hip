// DirichletProcessDiscrete::Shared -- the stick-breaking side of the model
// (include/distributions/models/dpd.hpp:59-101): which values exist, their
// betas, how many rows carry each, and the mass beta0 left for unseen values.
// Host side like the reference's (a Sparse_ map and a SparseCounter): a new
// value is DIST_UNLIKELY per row, and what the row update needs of it is the
// dense view (dist_shared_t: betas[dim], alpha, beta0) the kernels read.
//
// Dense slots: a value keeps its slot for life; a value whose last row left
// (remove_value -> count 0, dpd.hpp:78-83) leaves a slot with beta 0 that the
// next new value takes, so groups' count vectors never have to be re-packed.
//
// Entropy: sample_beta_safe (random.hpp:87-119) = two std::gamma_distribution
// <double> draws over rng_t.  The reference calls libstdc++'s <random>; so
// does this file, over an engine that IS std::minstd_rand0 (one word of
// state, random_fwd.hpp:34) -- tests/test_dpd_shared.py compares it with a
// probe that uses std::default_random_engine itself.
#include <algorithm>
#include <cstring>
#include <random>
#include <unordered_map>

#include "common.h"

namespace {

using dist::Error;

struct Engine {     // std::minstd_rand0 with its state in the caller's word
    typedef uint_fast32_t result_type;
    uint32_t * state;
    static constexpr result_type min() { return 1u; }
    static constexpr result_type max() { return 2147483646u; }
    result_type operator()() {
        *state = (uint32_t)(((uint64_t)*state * 16807ull) % 2147483647ull);
        return *state;
    }
};

float sample_gamma(Engine & rng, float alpha, float beta = 1.f) {
    std::gamma_distribution<double> sampler(alpha, beta);    // random.hpp:87-97
    return (float)sampler(rng);
}

float sample_beta(Engine & rng, float alpha, float beta) {  // random.hpp:99-108
    const float x = sample_gamma(rng, alpha);
    const float y = sample_gamma(rng, beta);
    if (x == 0 && y == 0) {
        std::uniform_real_distribution<float> sampler(0.0, 1.0);
        return sampler(rng) < alpha / (alpha + beta) ? 1.0f : 0.0f;
    }
    return x / (x + y);
}

float sample_beta_safe(Engine & rng, float alpha, float beta,
                       float min_value) {                  // random.hpp:110-119
    DIST_REQUIRE(min_value >= 0, "bad bound");
    DIST_REQUIRE(alpha > 0, "bad alpha");
    const float p = sample_beta(rng, alpha, beta);
    return (p + min_value) / (1.f + min_value);
}

constexpr float kMinBeta = 1e-6f;       // dpd.hpp:56 MIN_BETA()
constexpr uint32_t kDead = 0xFFFFFFFFu;

}  // namespace

struct dist_dpd_shared {
    float gamma = 1.f;
    float alpha = 1.f;
    float beta0 = 1.f;
    std::vector<float> betas;           // by dense slot (0 for a free slot)
    std::vector<uint32_t> values;       // by dense slot (kDead: free)
    std::vector<int> counts;            // by dense slot
    std::vector<uint32_t> free_slots;
    std::unordered_map<uint32_t, uint32_t> slot_of;
    uint64_t version = 0;               // bumps whenever the dense view moves

    void clear() {
        betas.clear(); values.clear(); counts.clear(); free_slots.clear();
        slot_of.clear();
        ++version;
    }
    uint32_t take_slot(uint32_t value, float beta, int count) {
        uint32_t slot;
        if (!free_slots.empty()) {
            slot = free_slots.back();
            free_slots.pop_back();
        } else {
            slot = (uint32_t)betas.size();
            betas.push_back(0.f);
            values.push_back(kDead);
            counts.push_back(0);
        }
        betas[slot] = beta;
        values[slot] = value;
        counts[slot] = count;
        slot_of[value] = slot;
        ++version;
        return slot;
    }
    void add_value(uint32_t value, Engine & rng) {          // dpd.hpp:66-74
        DIST_REQUIRE(value != DIST_DPD_OTHER, "cannot add OTHER");
        auto it = slot_of.find(value);
        if (it != slot_of.end()) {
            ++counts[it->second];
            return;
        }
        DIST_REQUIRE(beta0 > 0, "cannot add any more values");
        const float beta = beta0 * sample_beta_safe(rng, 1.f, gamma, kMinBeta);
        beta0 = std::max(kMinBeta, beta0 - beta);
        take_slot(value, beta, 1);
    }
    void remove_value(uint32_t value) {                     // dpd.hpp:76-83
        DIST_REQUIRE(value != DIST_DPD_OTHER, "cannot remove OTHER");
        auto it = slot_of.find(value);
        DIST_REQUIRE(it != slot_of.end(), "missing key");
        const uint32_t slot = it->second;
        if (--counts[slot] == 0) {
            beta0 = std::min(1.f, beta0 + betas[slot]);
            betas[slot] = 0.f;
            values[slot] = kDead;
            free_slots.push_back(slot);
            slot_of.erase(it);
            ++version;
        }
    }
    void realize(Engine & rng) {                            // dpd.hpp:85-101
        const size_t max_size = 10000;
        const float min_beta0 = 1e-4f;
        uint32_t new_value = 0;
        for (auto const & i : slot_of)
            new_value = std::max(new_value, 1 + i.first);
        while (slot_of.size() < max_size - 1 && beta0 > min_beta0)
            add_value(new_value++, rng);
        if (beta0 > 0) {
            add_value(new_value, rng);
            betas[slot_of[new_value]] += beta0;
            beta0 = 0;
            ++version;
        }
    }
};

extern "C" {

using dist::guarded;

int dist_sample_gamma(uint32_t * rng_state, float alpha, float beta,
                      float * out) {
    return guarded([&] {
        DIST_REQUIRE(alpha > 0 && beta > 0, "bad gamma parameters");
        Engine rng{rng_state};
        *out = sample_gamma(rng, alpha, beta);
    });
}

int dist_sample_beta_safe(uint32_t * rng_state, float alpha, float beta,
                          float min_value, float * out) {
    return guarded([&] {
        DIST_REQUIRE(beta > 0, "bad beta");
        Engine rng{rng_state};
        *out = sample_beta_safe(rng, alpha, beta, min_value);
    });
}

dist_dpd_shared_t * dist_dpd_shared_create(void) {
    try {
        return new dist_dpd_shared();
    } catch (...) {
        dist::set_last_error("out of memory");
        return nullptr;
    }
}

void dist_dpd_shared_destroy(dist_dpd_shared_t * s) { delete s; }

int dist_dpd_shared_copy(dist_dpd_shared_t * dst,
                         const dist_dpd_shared_t * src) {
    return guarded([&] {
        const uint64_t version = dst->version;
        *dst = *src;
        dst->version = version + 1;
    });
}

int dist_dpd_shared_load(dist_dpd_shared_t * s, float gamma, float alpha,
                         const uint32_t * values, const float * betas,
                         const int * counts, size_t n) {
    return guarded([&] {                                    // dpd.hpp:103-124
        double beta_sum = 0;
        for (size_t i = 0; i < n; ++i) {
            DIST_REQUIRE(values[i] != DIST_DPD_OTHER, "OTHER is not a value");
            DIST_REQUIRE(betas[i] > 0, "betas must be positive");
            beta_sum += betas[i];
        }
        DIST_REQUIRE(beta_sum <= 1 + 1e-4, "betas sum to more than 1");
        s->clear();
        s->gamma = gamma;
        s->alpha = alpha;
        for (size_t i = 0; i < n; ++i) {
            DIST_REQUIRE(!s->slot_of.count(values[i]), "duplicate key");
            s->take_slot(values[i], betas[i], counts ? counts[i] : 0);
        }
        s->beta0 = (float)std::max(0.0, 1.0 - beta_sum);
    });
}

int dist_dpd_shared_add_value(dist_dpd_shared_t * s, uint32_t value,
                              uint32_t * rng_state) {
    return guarded([&] {
        Engine rng{rng_state};
        s->add_value(value, rng);
    });
}

int dist_dpd_shared_remove_value(dist_dpd_shared_t * s, uint32_t value) {
    return guarded([&] { s->remove_value(value); });
}

int dist_dpd_shared_realize(dist_dpd_shared_t * s, uint32_t * rng_state) {
    return guarded([&] {
        Engine rng{rng_state};
        s->realize(rng);
    });
}

size_t dist_dpd_shared_slots(const dist_dpd_shared_t * s) {
    return s->betas.size();
}

size_t dist_dpd_shared_size(const dist_dpd_shared_t * s) {
    return s->slot_of.size();
}

uint64_t dist_dpd_shared_version(const dist_dpd_shared_t * s) {
    return s->version;
}

int dist_dpd_shared_params(const dist_dpd_shared_t * s, float * gamma,
                           float * alpha, float * beta0) {
    return guarded([&] {
        *gamma = s->gamma;
        *alpha = s->alpha;
        *beta0 = s->beta0;
    });
}

int dist_dpd_shared_view(const dist_dpd_shared_t * s, dist_shared_t * out) {
    return guarded([&] {
        memset(out, 0, sizeof(*out));
        out->kind = DIST_DPD;
        out->dim = (int)s->betas.size();
        out->p[0] = s->alpha;
        out->p[1] = s->beta0;
        out->betas = s->betas.empty() ? nullptr : s->betas.data();
    });
}

int dist_dpd_shared_slot(const dist_dpd_shared_t * s, uint32_t value,
                         uint32_t * slot_out) {
    return guarded([&] {
        if (value == DIST_DPD_OTHER) {
            *slot_out = DIST_DPD_OTHER;
            return;
        }
        auto it = s->slot_of.find(value);
        DIST_REQUIRE(it != s->slot_of.end(), "unknown value");
        *slot_out = it->second;
    });
}

int dist_dpd_shared_dump(const dist_dpd_shared_t * s, uint32_t * values,
                         float * betas, int * counts) {
    return guarded([&] {
        for (size_t i = 0; i < s->betas.size(); ++i) {
            values[i] = s->values[i];
            betas[i] = s->betas[i];
            counts[i] = s->counts[i];
        }
    });
}

}  // extern "C"
