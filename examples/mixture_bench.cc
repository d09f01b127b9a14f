// The reference's mixture benchmark loop (benchmarks/mixture.cc:79-115) as a
// downstream program writes it against the reference's headers: the includes,
// the namespace, Model::Shared::EXAMPLE(), mixture.groups(), the remove /
// score_value (accumulating) / add loop and its "cells/us" figure.  It builds
// against THIS library by putting include/compat first on the include path:
//   g++ -std=c++11 -Iinclude/compat examples/mixture_bench.cc
//       -Ldistributions_amd -ldistributions_hip -Wl,-rpath,$PWD/distributions_amd
// (Group::sample_value, which the reference's benchmark draws its values with,
// belongs to the samplers -- out of this library's scope -- so values come
// from sample_int / sample_unif01 here; the Scorers half of the benchmark,
// mixture.cc:119-131, times the per-group Scorer objects, which the batched
// engine has no use for.)
#include <iomanip>
#include <iostream>
#include <typeinfo>
#include <vector>

#include <distributions/vector.hpp>
#include <distributions/models/bb.hpp>
#include <distributions/models/dd.hpp>
#include <distributions/models/gp.hpp>
#include <distributions/models/bnb.hpp>
#include <distributions/models/nich.hpp>
#include <distributions/timers.hpp>

using namespace distributions;  // NOLINT(*)

rng_t rng;

template <class Value> Value draw_value();
template <> int draw_value<int>() { return sample_int(rng, 0, 3); }
template <> bool draw_value<bool>() { return sample_int(rng, 0, 1) != 0; }
template <> uint32_t draw_value<uint32_t>() {
    return (uint32_t)sample_int(rng, 0, 12);
}
template <> float draw_value<float>() { return 6.f * sample_unif01(rng) - 3.f; }

template <class Model>
double speedtest(const typename Model::Shared & shared, size_t group_count,
                 size_t iters, double * checksum) {
    typename Model::Mixture mixture;
    mixture.groups().resize(group_count);
    std::vector<typename Model::Value> values;
    std::vector<size_t> assignments;
    for (size_t groupid = 0; groupid < group_count; ++groupid) {
        typename Model::Group & group = mixture.groups()[groupid];
        group.init(shared, rng);
    }
    for (size_t i = 0; i < 4 * group_count; ++i) {
        size_t groupid = sample_int(rng, 0, group_count - 1);
        typename Model::Group & group = mixture.groups()[groupid];
        typename Model::Value value = draw_value<typename Model::Value>();
        group.add_value(shared, value, rng);
        values.push_back(value);
        assignments.push_back(groupid);
    }
    mixture.init(shared, rng);
    VectorFloat scores(group_count);

    int64_t time = -current_time_us();
    for (size_t i = 0; i < iters / 8; ++i) {
        vector_zero(scores.size(), scores.data());
        for (size_t j = 0; j < 8; ++j) {
            size_t k = (8 * i + j) % values.size();
            typename Model::Value value = values[k];
            size_t groupid = assignments[k];
            mixture.remove_value(shared, groupid, value, rng);
            mixture.score_value(shared, value, scores, rng);
            mixture.add_value(shared, groupid, value, rng);
        }
    }
    time += current_time_us();
    for (size_t g = 0; g < group_count; ++g) *checksum += scores[g];
    return iters * 1e0 / time;
}

template <class Model>
void speedtests(size_t max_groups) {
    std::cout << demangle(typeid(typename Model::Shared).name()) << '\n'
              << "Groups" << '\t' << "Mixture (cells/us)" << '\n';
    auto const shared = Model::Shared::EXAMPLE();
    double checksum = 0;
    for (size_t group_count = 1; group_count <= max_groups; group_count *= 10) {
        size_t iters = 8 * (200 / (1 + group_count / 100) + 1);
        double rate = speedtest<Model>(shared, group_count, iters, &checksum);
        std::cout << group_count << '\t' << std::right << std::setw(7)
                  << std::fixed << std::setprecision(4) << rate << '\n';
    }
    std::cout << "checksum " << std::setprecision(6) << checksum << '\n';
}

int main(int argc, char ** argv) {
    const size_t max_groups = argc > 1 ? (size_t)atoi(argv[1]) : 1000;
    speedtests<BetaBernoulli>(max_groups);
    speedtests<DirichletDiscrete<4>>(max_groups);
    speedtests<GammaPoisson>(max_groups);
    speedtests<BetaNegativeBinomial>(max_groups);
    speedtests<NormalInverseChiSq>(max_groups);
    return 0;
}
