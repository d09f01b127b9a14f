// distributions_hip.hpp -- header shim: the reference's C++ class surface for
// the mixture row-update path, forwarding to the C ABI of
// libdistributions_hip.so (distributions_hip.h).
//
// A downstream program that instantiates `Model::Mixture` /
// `Clustering<int>::PitmanYor::Mixture` from the reference's headers
// (include/distributions/mixture.hpp:340-450, clustering.hpp:126-234) can
// switch to these classes by changing the namespace; member names, argument
// order, the accumulate/overwrite semantics of score_value and the
// add_value/remove_value return flags are the reference's.  Differences, all
// forced by the state living in HBM:
//   - `rng_t &` arguments are accepted and ignored where the reference ignores
//     them too (every Mixture member on this path);
//   - groups(i) returns a copy (like the lp wrapper's Mixture.__getitem__,
//     lp/models/_dd.pyx:96-100), not a reference;
//   - errors throw std::runtime_error (the reference does under
//     DIST_THROW_ON_ERROR, common.hpp:49-57).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "distributions_hip.h"

namespace distributions_hip {

// rng_t (random_fwd.hpp:34: std::default_random_engine == minstd_rand0 in
// libstdc++): the engine state, advanced by the library (dist_rng_next).  A
// default-constructed engine is seeded with 1 like the reference's; it also
// satisfies UniformRandomBitGenerator, so <random> distributions drawing from
// it give what they give over std::minstd_rand0.
// `rng_t rng(seed)` SEEDS the engine, as std::default_random_engine(seed) does
// (the seed reduced mod 2^31 - 1, 0 mapped to 1); a raw engine state -- what
// dist_rng_seed / dist_rng_jump return -- goes through rng_t::from_state().
struct rng_t {
    typedef uint32_t result_type;
    uint32_t state;
    rng_t() : state(1u) {}
    explicit rng_t(uint32_t seed_value) : state(dist_rng_seed(seed_value)) {}
    static rng_t from_state(uint32_t engine_state) {
        rng_t r;
        r.state = engine_state;
        return r;
    }
    void seed(uint32_t value) { state = dist_rng_seed(value); }
    static constexpr result_type min() { return 1u; }
    static constexpr result_type max() { return 2147483646u; }
    result_type operator()() { return dist_rng_next(&state); }
};

inline void check(int rc) {
    if (rc != 0) throw std::runtime_error(dist_last_error());
}

// AlignedFloats / VectorFloat stand-in (vector.hpp:63-90)
typedef std::vector<float> VectorFloat;

namespace detail {
inline uint32_t word(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
inline uint32_t word(int x) { return (uint32_t)x; }
inline uint32_t word(uint32_t x) { return x; }
inline uint32_t word(bool x) { return x ? 1u : 0u; }
}  // namespace detail

// Model::{Shared, Group, Mixture} over one dist_kind.
template <int KIND, class Value_>
struct Model {
    typedef Value_ Value;

    struct Shared : dist_shared_t {
        Shared() { memset(static_cast<dist_shared_t *>(this), 0,
                          sizeof(dist_shared_t)); kind = KIND; }
        // Shared::EXAMPLE() of the reference's models (dd.hpp:78-85 with
        // `example_dim` categories, bb.hpp:70-75, gp.hpp:75-80, nich.hpp:87-94,
        // bnb.hpp:79-85); DirichletProcessDiscrete's (dpd.hpp:141-152) needs
        // storage for its betas: set dim / betas yourself
        static Shared EXAMPLE(int example_dim = DIST_DD_MAX_DIM) {
            Shared shared;
            switch (KIND) {
            case DIST_DD:
                shared.dim = example_dim;
                for (int i = 0; i < example_dim; ++i) shared.alphas[i] = 0.5f;
                break;
            case DIST_BB: shared.p[0] = 0.5f; shared.p[1] = 2.0f; break;
            case DIST_GP: shared.p[0] = 1.0f; shared.p[1] = 1.0f; break;
            case DIST_NICH:
                shared.p[0] = 0.0f; shared.p[1] = 1.0f;
                shared.p[2] = 1.0f; shared.p[3] = 1.0f;
                break;
            case DIST_BNB:
                shared.p[0] = 1.0f; shared.p[1] = 1.0f; shared.p[2] = 1.0f;
                break;
            default: break;
            }
            return shared;
        }
    };

    struct Group {
        std::vector<uint32_t> words;
        void init(const Shared & shared, rng_t &) {
            words.assign(dist_group_words(&shared), 0);
            check(dist_group_init(&shared, words.data()));
        }
        void add_value(const Shared & shared, const Value & value, rng_t &) {
            check(dist_group_add_value(&shared, words.data(),
                                       detail::word(value)));
        }
        void remove_value(const Shared & shared, const Value & value, rng_t &) {
            check(dist_group_remove_value(&shared, words.data(),
                                          detail::word(value)));
        }
        // the Model::Group message of schema.proto as wire bytes (what
        // protobuf_dump/protobuf_load exchange via generated classes in the
        // reference, dd.hpp:94-111 etc.); keys: DirichletProcessDiscrete's
        // dense index -> value table, else nullptr
        std::string protobuf_dump(const Shared & shared,
                                  const uint32_t * keys = nullptr) const {
            size_t n = 0;
            check(dist_group_protobuf_dump(&shared, words.data(), keys,
                                           nullptr, 0, &n));
            std::string out(n, '\0');
            check(dist_group_protobuf_dump(
                &shared, words.data(), keys,
                reinterpret_cast<uint8_t *>(&out[0]), n, &n));
            return out;
        }
        void protobuf_load(const Shared & shared, const std::string & data,
                           const uint32_t * keys = nullptr) {
            words.assign(dist_group_words(&shared), 0);
            check(dist_group_protobuf_load(
                &shared, keys, reinterpret_cast<const uint8_t *>(data.data()),
                data.size(), words.data()));
        }
        float score_value(const Shared & shared, const Value & value,
                          rng_t &) const {
            float out = 0;
            check(dist_group_score_value(&shared, words.data(),
                                         detail::word(value), &out));
            return out;
        }
        // Group::sample_value (dd.hpp:188-199, bb.hpp:154-160, gp.hpp:166-172,
        // nich.hpp:204-210, bnb.hpp:168-174): a draw from the posterior
        // predictive through the model's Sampler (init, then eval), on the
        // host with the <random> distributions random.hpp:42-108 uses, over
        // rng_t.  A sampler, not part of the accelerated path: what a caller
        // like benchmarks/mixture.cc:94 draws its data with.
        Value sample_value(const Shared & shared, rng_t & rng) const {
            return sample_value_impl(shared, rng, static_cast<Value *>(nullptr));
        }

      private:
        static float gamma_(rng_t & rng, float alpha, float beta = 1.f) {
            std::gamma_distribution<double> d(alpha, beta);   // random.hpp:87-97
            return (float)d(rng);
        }
        static float word_f(uint32_t w) { float f; memcpy(&f, &w, 4); return f; }
        template <class V>
        V sample_value_impl(const Shared & shared, rng_t & rng, V *) const {
            switch (KIND) {
            case DIST_DD: {   // Sampler: dirichlet(alphas + counts), discrete
                float ps[DIST_DD_MAX_DIM];
                float total = 0.f;
                for (int v = 0; v < shared.dim; ++v) {
                    const float a = shared.alphas[v] + (float)(int)words[1 + v];
                    ps[v] = a > 0 ? gamma_(rng, a) : 0.f;   // random.cc:121-137
                    total += ps[v];
                }
                const float scale = 1.f / total;
                float t = dist_rng_unif01(&rng.state);      // random.hpp:300-313
                for (int v = 0; v + 1 < shared.dim; ++v) {
                    t -= ps[v] * scale;
                    if (t < 0) return (V)v;
                }
                return (V)(shared.dim - 1);
            }
            case DIST_DPD: {   // dpd.hpp:275-305 (values 0 .. dim-1, then OTHER)
                std::vector<float> ps;
                for (int v = 0; v < shared.dim; ++v)
                    ps.push_back(shared.betas[v] * shared.p[0]
                                 + (float)(int)words[1 + v]);
                if (shared.p[1] > 0) ps.push_back(shared.p[1] * shared.p[0]);
                float total = 0.f;
                for (float & q : ps) {
                    q = q > 0 ? gamma_(rng, q) : 0.f;       // random.cc:121-137
                    total += q;
                }
                const float scale = 1.f / total;
                float t = dist_rng_unif01(&rng.state);      // random.hpp:300-313
                size_t index = ps.size() - 1;
                for (size_t i = 0; i + 1 < ps.size(); ++i) {
                    t -= ps[i] * scale;
                    if (t < 0) { index = i; break; }
                }
                return index < (size_t)shared.dim ? (V)index
                                                  : (V)DIST_DPD_OTHER;
            }
            case DIST_BB: {
                const float x = gamma_(rng, shared.p[0] + (float)(int)words[0]);
                const float y = gamma_(rng, shared.p[1] + (float)(int)words[1]);
                const float heads = x * (1.f / (x + y));
                return (V)(dist_rng_unif01(&rng.state) < heads);
            }
            case DIST_GP: {
                const float a = shared.p[0] + (float)words[1];
                const float inv_b = shared.p[1] + (float)words[0];
                std::poisson_distribution<int> d(gamma_(rng, a, 1.f / inv_b));
                return (V)d(rng);
            }
            case DIST_BNB: {   // bnb.hpp:177-192
                const float r = shared.p[2];
                const float a = shared.p[0] + r * (float)words[0];
                const float b = shared.p[1] + (float)words[1];
                const float x = gamma_(rng, a), y = gamma_(rng, b);
                float beta = x / (x + y);
                if (x == 0 && y == 0)
                    beta = dist_rng_unif01(&rng.state) < a / (a + b) ? 1.f : 0.f;
                std::negative_binomial_distribution<int> d((int)r, beta);
                return (V)d(rng);
            }
            case DIST_NICH: {   // nich.hpp:58-69, 213-231
                const float mu = shared.p[0], kappa = shared.p[1];
                const float sigmasq = shared.p[2], nu = shared.p[3];
                const float n = (float)(int)words[0];
                const float mean = word_f(words[1]), ctv = word_f(words[2]);
                const float mu_1 = mu - mean;
                const float pk = kappa + n;
                const float pmu = (kappa * mu + mean * n) / pk;
                const float pnu = nu + n;
                const float psig = 1.f / pnu * (nu * sigmasq + ctv
                                                + (n * kappa * mu_1 * mu_1) / pk);
                std::chi_squared_distribution<double> chi(pnu);
                const float s2 = pnu * psig / (float)chi(rng);
                std::normal_distribution<float> m(pmu, sqrtf(s2 / pk));
                const float centre = m(rng);
                std::normal_distribution<float> d(centre, sqrtf(s2));
                return (V)d(rng);
            }
            default:
                throw std::runtime_error("sample_value: no sampler for this "
                                         "model in the shim");
            }
        }

      public:
    };

    // Model::Scorer (dd.hpp:222-245, bb.hpp:185-205, gp.hpp:198-217,
    // nich.hpp:239-259, bnb.hpp:195-223, dpd.hpp:309-341): one group's
    // scorer, the per-group counterpart of Mixture::score_value -- what
    // benchmarks/mixture.cc:41-74 keeps beside the mixture and times
    // against it.  Host arithmetic (dist_scorer_init / dist_scorer_eval).
    struct Scorer {
        std::vector<float> state;
        void init(const Shared & shared, const Group & group, rng_t &) {
            state.assign(dist_scorer_words(&shared), 0.f);
            check(dist_scorer_init(&shared, group.words.data(), state.data()));
        }
        float eval(const Shared & shared, const Value & value, rng_t &) const {
            float out = 0;
            check(dist_scorer_eval(&shared, state.data(), detail::word(value),
                                   &out));
            return out;
        }
    };

    // MixtureSlave<Model, ...> (mixture.hpp:340-450)
    class Mixture {
      public:
        Mixture() : ptr_(nullptr), handed_over_(false) {}
        ~Mixture() { if (ptr_) dist_mixture_destroy(ptr_); }
        Mixture(const Mixture &) = delete;
        Mixture & operator=(const Mixture &) = delete;

        // groups().resize(n) / groups()[i] / groups().push_back(group)
        // BEFORE init(), as the reference's callers fill a mixture
        // (benchmarks/mixture.cc:84-100): host-side groups that init() hands
        // to the device.  They stay readable afterwards -- the snapshot init()
        // took, as benchmarks/mixture.cc:55-66 copies them out of a `const
        // Mixture &` -- while the LIVE statistics are in HBM:
        // groups(shared, i) returns a copy of those.  Taking the mutable
        // reference again tells init() to hand the groups over anew.
        std::vector<Group> & groups() { handed_over_ = false; return staged_; }
        const std::vector<Group> & groups() const { return staged_; }
        void append(const Shared & shared, const Group & group) {
            check(dist_mixture_append(handle(shared), group.words.data()));
        }
        size_t size() const { return ptr_ ? dist_mixture_size(ptr_) : 0; }
        Group groups(const Shared & shared, size_t i) {
            Group g;
            g.words.assign(dist_group_words(&shared), 0);
            check(dist_mixture_get_group(handle(shared), i, g.words.data()));
            return g;
        }
        void init(const Shared & shared, rng_t &) {
            if (!staged_.empty() && !handed_over_) {
                check(dist_mixture_clear(handle(shared)));
                for (const Group & group : staged_) append(shared, group);
                handed_over_ = true;
            }
            check(dist_mixture_init(handle(shared)));
        }
        void add_group(const Shared & shared, rng_t &) {
            check(dist_mixture_add_group(handle(shared)));
        }
        void remove_group(const Shared & shared, size_t groupid) {
            check(dist_mixture_remove_group(handle(shared), groupid));
        }
        void add_value(const Shared & shared, size_t groupid,
                       const Value & value, rng_t &) {
            check(dist_mixture_add_value(handle(shared), groupid,
                                         detail::word(value)));
        }
        void remove_value(const Shared & shared, size_t groupid,
                          const Value & value, rng_t &) {
            check(dist_mixture_remove_value(handle(shared), groupid,
                                            detail::word(value)));
        }
        float score_value_group(const Shared & shared, size_t groupid,
                                const Value & value, rng_t &) {
            float out = 0;
            check(dist_mixture_score_value_group(handle(shared), groupid,
                                                 detail::word(value), &out));
            return out;
        }
        // accumulates into scores_accum (mixture.hpp:416-425)
        void score_value(const Shared & shared, const Value & value,
                         VectorFloat & scores_accum, rng_t &) {
            check(dist_mixture_score_value(handle(shared), detail::word(value),
                                           scores_accum.data(),
                                           scores_accum.size()));
        }
        // score_value for a batch of values in one launch (extension):
        // scores_accum[r * size() + k] accumulates the score of values[r] in
        // group k -- what values.size() calls of score_value leave, without a
        // launch and a round trip per value
        void score_values(const Shared & shared,
                          const std::vector<Value> & values,
                          VectorFloat & scores_accum, rng_t &) {
            const size_t k = dist_mixture_size(handle(shared));
            if (scores_accum.size() != values.size() * k)
                throw std::invalid_argument(
                    "scores_accum.size() != values.size() * size()");
            std::vector<uint32_t> words(values.size());
            for (size_t i = 0; i < values.size(); ++i)
                words[i] = detail::word(values[i]);
            check(dist_mixture_score_values(handle(shared), words.data(),
                                            words.size(), scores_accum.data(),
                                            k));
        }
        // mixture.hpp:427-431
        float score_data(const Shared & shared, rng_t &) {
            float out = 0;
            check(dist_mixture_score_data(handle(shared), &out));
            return out;
        }
        // mixture.hpp:433-438: one score per candidate Shared
        template <class SharedT>   // Shared, or a model's own (EXAMPLE-only)
        void score_data_grid(const std::vector<SharedT> & shareds,
                             VectorFloat & scores_out, rng_t &) {
            static_assert(sizeof(SharedT) == sizeof(dist_shared_t),
                          "Shared adds no members");
            if (shareds.size() != scores_out.size())
                throw std::invalid_argument("shareds.size() != scores_out.size()");
            if (shareds.empty()) return;
            check(dist_mixture_score_data_grid(
                handle(shareds[0]),
                reinterpret_cast<const dist_shared_t *>(shareds.data()),
                shareds.size(), scores_out.data()));
        }

      private:
        dist_mixture_t * handle(const Shared & shared) {
            if (!ptr_) {
                ptr_ = dist_mixture_create(&shared);
                if (!ptr_) throw std::runtime_error(dist_last_error());
            }
            return ptr_;
        }
        dist_mixture_t * ptr_;
        std::vector<Group> staged_;
        bool handed_over_;
    };
};

// models/dd.hpp:41-54: DirichletDiscrete<max_dim>; the library holds up to
// 256 categories whatever max_dim says, which only sizes EXAMPLE()
template <int max_dim_ = DIST_DD_MAX_DIM>
struct DirichletDiscrete : Model<DIST_DD, int> {
    static_assert(max_dim_ >= 1 && max_dim_ <= DIST_DD_MAX_DIM,
                  "1 <= max_dim <= 256");
    enum { max_dim = max_dim_ };
    typedef Model<DIST_DD, int> Base;
    struct Shared : Base::Shared {
        static Shared EXAMPLE() {
            Shared shared;
            static_cast<Base::Shared &>(shared) = Base::Shared::EXAMPLE(max_dim);
            return shared;
        }
    };
};
typedef Model<DIST_BB, bool> BetaBernoulli;              // models/bb.hpp
typedef Model<DIST_GP, uint32_t> GammaPoisson;           // models/gp.hpp
typedef Model<DIST_NICH, float> NormalInverseChiSq;      // models/nich.hpp
// models/dpd.hpp: the Shared owns its betas (values 0 .. dim-1 of the dense
// remap; dist_shared_t carries a pointer to them)
struct DirichletProcessDiscrete : Model<DIST_DPD, uint32_t> {
    typedef Model<DIST_DPD, uint32_t> Base;
    // dpd.hpp:59-153.  The members a mixture reads (dim, betas by DENSE SLOT,
    // p[0] = alpha, p[1] = beta0) are the base's; the stick-breaking state
    // behind them -- which value owns which slot, the row counts, gamma --
    // is a dist_dpd_shared_t.  Groups and mixtures count values under their
    // slot: pass shared.slot(value) where the reference passes the value
    // (for values 0..V-1 loaded in order, e.g. EXAMPLE(), the two coincide).
    struct Shared : Base::Shared {
        Shared() : handle(dist_dpd_shared_create()) {
            if (!handle) throw std::runtime_error(dist_last_error());
            refresh();
        }
        Shared(const Shared & other) : Base::Shared(other),
                                       handle(dist_dpd_shared_create()) {
            if (!handle) throw std::runtime_error(dist_last_error());
            copy_state(other);
        }
        Shared & operator=(const Shared & other) {
            if (this != &other) copy_state(other);
            return *this;
        }
        ~Shared() { dist_dpd_shared_destroy(handle); }

        float gamma() const { return scalar(0); }
        float alpha() const { return scalar(1); }
        float beta0() const { return scalar(2); }
        size_t size() const { return dist_dpd_shared_size(handle); }

        // protobuf_load's content (dpd.hpp:103-124)
        void load(float gamma_value, float alpha_value,
                  const std::vector<uint32_t> & values,
                  const std::vector<float> & betas_by_value,
                  const std::vector<int> & counts) {
            check(dist_dpd_shared_load(handle, gamma_value, alpha_value,
                                       values.data(), betas_by_value.data(),
                                       counts.empty() ? nullptr : counts.data(),
                                       values.size()));
            refresh();
        }
        // values 0 .. n-1 with these betas (the dense layout itself)
        void set_betas(const std::vector<float> & values) {
            std::vector<uint32_t> keys(values.size());
            for (size_t i = 0; i < keys.size(); ++i) keys[i] = (uint32_t)i;
            const float a = this->p[0];
            load(gamma(), a, keys, values, std::vector<int>(values.size(), 1));
        }
        void add_value(const Value & value, rng_t & rng) {   // dpd.hpp:66-74
            check(dist_dpd_shared_add_value(handle, value, &rng.state));
            refresh();
        }
        void remove_value(const Value & value, rng_t &) {    // dpd.hpp:76-83
            check(dist_dpd_shared_remove_value(handle, value));
            refresh();
        }
        void realize(rng_t & rng) {                          // dpd.hpp:85-101
            check(dist_dpd_shared_realize(handle, &rng.state));
            refresh();
        }
        // the dense slot a value is counted under (OTHER stays OTHER)
        uint32_t slot(const Value & value) const {
            uint32_t out = 0;
            check(dist_dpd_shared_slot(handle, value, &out));
            return out;
        }
        // dpd.hpp:141-152: alpha 0.5, beta0 0, a hundred values of 1/100
        static Shared EXAMPLE() {
            Shared shared;
            std::vector<uint32_t> keys(100);
            for (size_t i = 0; i < keys.size(); ++i) keys[i] = (uint32_t)i;
            shared.load((float)(1.0 / 100), 0.5f, keys,
                        std::vector<float>(100, (float)(1.0 / 100)),
                        std::vector<int>(100, 1));
            shared.p[1] = 0.0f;     // "must be zero for testing"
            return shared;
        }

      private:
        dist_dpd_shared_t * handle;
        float scalar(int which) const {
            float v[3];
            check(dist_dpd_shared_params(handle, &v[0], &v[1], &v[2]));
            return v[which];
        }
        void refresh() {
            dist_shared_t view;
            check(dist_dpd_shared_view(handle, &view));
            static_cast<dist_shared_t &>(*this) = view;
        }
        void copy_state(const Shared & other) {
            check(dist_dpd_shared_copy(handle, other.handle));
            refresh();
            this->p[1] = other.p[1];
        }
    };
};
typedef Model<DIST_BNB, uint32_t> BetaNegativeBinomial;      // models/bnb.hpp

// Clustering<int>::PitmanYor (clustering.hpp:58-234)
struct PitmanYor {
    float alpha;
    float d;

    float score_add_value(int group_size, int nonempty_group_count,
                          int sample_size, int empty_group_count = 1) const {
        float out = 0;
        check(dist_py_score_add_value(alpha, d, group_size,
                                      nonempty_group_count, sample_size,
                                      empty_group_count, &out));
        return out;
    }
    float score_remove_value(int group_size, int nonempty_group_count,
                             int sample_size, int empty_group_count = 1) const {
        float out = 0;
        check(dist_py_score_remove_value(alpha, d, group_size,
                                         nonempty_group_count, sample_size,
                                         empty_group_count, &out));
        return out;
    }

    class Mixture {   // CachedMixture
      public:
        Mixture() : ptr_(dist_py_mixture_create()) {
            if (!ptr_) throw std::runtime_error(dist_last_error());
        }
        ~Mixture() { dist_py_mixture_destroy(ptr_); }
        Mixture(const Mixture &) = delete;
        Mixture & operator=(const Mixture &) = delete;

        std::vector<int> & counts() { return staged_; }   // set before init()
        void init(const PitmanYor & model) {
            check(dist_py_mixture_init(ptr_, model.alpha, model.d,
                                       staged_.data(), staged_.size()));
        }
        bool add_value(const PitmanYor & model, size_t groupid) {
            int flag = 0;
            check(dist_py_mixture_add_value(ptr_, model.alpha, model.d,
                                            groupid, &flag));
            return flag != 0;
        }
        bool remove_value(const PitmanYor & model, size_t groupid) {
            int flag = 0;
            check(dist_py_mixture_remove_value(ptr_, model.alpha, model.d,
                                               groupid, &flag));
            return flag != 0;
        }
        // overwrites scores (clustering.hpp:195-208)
        void score_value(const PitmanYor & model, VectorFloat & scores) const {
            check(dist_py_mixture_score_value(ptr_, model.alpha, model.d,
                                              scores.data(), scores.size()));
        }
        size_t size() const { return dist_py_mixture_size(ptr_); }
        size_t sample_size() const { return dist_py_mixture_sample_size(ptr_); }
        std::vector<size_t> empty_groupids() const {
            std::vector<size_t> ids(dist_py_mixture_empty_groupids(ptr_, nullptr, 0));
            dist_py_mixture_empty_groupids(ptr_, ids.data(), ids.size());
            return ids;
        }

      private:
        dist_py_mixture_t * ptr_;
        std::vector<int> staged_;
    };
};

// MixtureIdTracker (mixture.hpp:460-521)
class MixtureIdTracker {
  public:
    typedef uint32_t Id;
    MixtureIdTracker() : ptr_(dist_id_tracker_create()) {}
    ~MixtureIdTracker() { dist_id_tracker_destroy(ptr_); }
    MixtureIdTracker(const MixtureIdTracker &) = delete;
    MixtureIdTracker & operator=(const MixtureIdTracker &) = delete;
    void init(size_t group_count = 0) { check(dist_id_tracker_init(ptr_, group_count)); }
    void add_group() { check(dist_id_tracker_add_group(ptr_)); }
    void remove_group(Id packed) { check(dist_id_tracker_remove_group(ptr_, packed)); }
    Id packed_to_global(Id packed) const {
        Id out = 0;
        check(dist_id_tracker_packed_to_global(ptr_, packed, &out));
        return out;
    }
    Id global_to_packed(Id global) const {
        Id out = 0;
        check(dist_id_tracker_global_to_packed(ptr_, global, &out));
        return out;
    }
    size_t packed_size() const { return dist_id_tracker_packed_size(ptr_); }
    size_t global_size() const { return dist_id_tracker_global_size(ptr_); }

  private:
    dist_id_tracker_t * ptr_;
};

// sample_from_scores_overwrite (random.hpp:361-366)
inline size_t sample_from_scores_overwrite(rng_t & rng, VectorFloat & scores) {
    size_t sample = 0;
    check(dist_sample_from_scores_overwrite(&rng.state, scores.size(),
                                            scores.data(), &sample));
    return sample;
}

}  // namespace distributions_hip

// ---------------------------------------------------------------------------
// Opt-in: the reference's own names.  With DISTRIBUTIONS_HIP_AS_DISTRIBUTIONS
// defined (the forwarding headers under include/compat/distributions/ define
// it), `namespace distributions` holds everything above plus the templates and
// helpers a caller of the reference's headers uses around the mixture path --
// sample_int / sample_unif01 (random.hpp:42-50), vector_zero (vector_math.hpp:31), current_time_us
// (timers.hpp:35), demangle (common.hpp:122) -- so that a file written against
// benchmarks/mixture.cc:79-115 compiles with only its include path changed.
#ifdef DISTRIBUTIONS_HIP_AS_DISTRIBUTIONS
#include <cxxabi.h>
#include <sys/time.h>

#include <random>

namespace distributions {
using namespace distributions_hip;   // NOLINT

inline int sample_int(rng_t & rng, int low, int high) {
    std::uniform_int_distribution<> sampler(low, high);
    return sampler(rng);
}
inline float sample_unif01(rng_t & rng) { return dist_rng_unif01(&rng.state); }
inline void vector_zero(size_t size, float * data) {
    for (size_t i = 0; i < size; ++i) data[i] = 0.f;
}
inline int64_t current_time_us() {
    timeval t;
    gettimeofday(&t, nullptr);
    return (int64_t)t.tv_usec + 1000000LL * (int64_t)t.tv_sec;
}
inline std::string demangle(const char * name) {
    int status = 0;
    char * text = abi::__cxa_demangle(name, nullptr, nullptr, &status);
    std::string out = (status == 0 && text) ? text : name;
    free(text);
    return out;
}
}  // namespace distributions
#endif  // DISTRIBUTIONS_HIP_AS_DISTRIBUTIONS
