// Model::Scorer against Mixture::score_value_group, model by model: the
// per-group scorer (benchmarks/mixture.cc:41-74) and the mixture's cached
// one must give the same score for the same (group, value) -- the check of
// distributions/tests/test_models.py:537-594 (test_mixture_score), through
// the reference's C++ names.  Prints, per model, the pairs compared, how many
// were bit-identical and the largest difference.
//   g++ -std=c++11 -Iinclude/compat examples/scorer_check.cc
//       -Ldistributions_amd -ldistributions_hip -o examples/scorer_check
#include <cmath>
#include <cstdio>
#include <cstring>
#include <typeinfo>
#include <vector>

#include <distributions/models/bb.hpp>
#include <distributions/models/bnb.hpp>
#include <distributions/models/dd.hpp>
#include <distributions/models/dpd.hpp>
#include <distributions/models/gp.hpp>
#include <distributions/models/nich.hpp>
#include <distributions/random.hpp>

using namespace distributions;  // NOLINT(*)

template <class Model>
void check(const char * name, size_t group_count) {
    rng_t rng;
    auto const shared = Model::Shared::EXAMPLE();
    typename Model::Mixture mixture;
    mixture.groups().resize(group_count);
    std::vector<typename Model::Value> values;
    for (size_t g = 0; g < group_count; ++g) {
        typename Model::Group & group = mixture.groups()[g];
        group.init(shared, rng);
        for (size_t i = 0; i < 3 * g + 1; ++i) {   // groups of every size
            typename Model::Value value = group.sample_value(shared, rng);
            group.add_value(shared, value, rng);
            values.push_back(value);
        }
    }
    mixture.init(shared, rng);
    const typename Model::Mixture & frozen = mixture;
    size_t pairs = 0, same = 0;
    double worst = 0;
    for (size_t g = 0; g < group_count; ++g) {
        typename Model::Scorer scorer;
        scorer.init(shared, frozen.groups()[g], rng);
        for (size_t i = 0; i < values.size(); i += 7) {
            const float a = scorer.eval(shared, values[i], rng);
            const float b = mixture.score_value_group(shared, g, values[i], rng);
            const float c = frozen.groups()[g].score_value(shared, values[i], rng);
            pairs += 1;
            same += std::memcmp(&a, &b, 4) == 0;
            // (Group::score_value IS Scorer init + eval, dd.hpp:160-167 etc.)
            if (std::memcmp(&a, &c, 4) != 0) worst = 1e30;
            const double diff = std::fabs((double)a - b)
                              / (1.0 + std::fabs((double)b));
            if (diff > worst) worst = diff;
        }
    }
    std::printf("%s pairs %zu identical %zu worst %.3g\n", name, pairs, same,
                worst);
}

int main() {
    check<BetaBernoulli>("bb", 12);
    check<DirichletDiscrete<16>>("dd", 12);
    check<DirichletProcessDiscrete>("dpd", 12);
    check<GammaPoisson>("gp", 12);
    check<BetaNegativeBinomial>("bnb", 12);
    check<NormalInverseChiSq>("nich", 12);
    return 0;
}
