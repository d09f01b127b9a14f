// Remove / score / add churn through the Mixture interface, written against
// the reference's header names so that it builds unchanged against either
// library: put this library's include/compat first on the include path,
//   g++ -std=c++11 -Iinclude/compat examples/mixture_bench.cc
//       -Ldistributions_amd -ldistributions_hip -Wl,-rpath,$PWD/distributions_amd
// For each component model it plants a table -- every group draws its rows
// from its own posterior predictive (Group::sample_value) -- and then times
// the three calls a Gibbs row update makes on a feature's mixture (take the
// row out of its group, score it against every group, put it back), reporting
// (row, group) cells per microsecond for 1, 10, 100, ... groups, plus a
// checksum of the last score vectors so that the work cannot be optimised
// away and two libraries can be compared.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <typeinfo>
#include <vector>

#include <distributions/vector.hpp>
#include <distributions/models/bb.hpp>
#include <distributions/models/dd.hpp>
#include <distributions/models/gp.hpp>
#include <distributions/models/bnb.hpp>
#include <distributions/models/nich.hpp>
#include <distributions/timers.hpp>

namespace {

using distributions::rng_t;
using distributions::VectorFloat;

// a planted table of one feature: rows, the group each sits in, the mixture
template <class M>
class Table {
  public:
    typedef typename M::Value Value;

    Table(const typename M::Shared & shared, size_t groups, size_t rows_each,
          rng_t & rng)
        : shared_(shared), scores_(groups, 0.f) {
        mixture_.groups().resize(groups);
        for (auto & group : mixture_.groups()) group.init(shared_, rng);
        // round-robin over the groups: a group's next row comes from its own
        // posterior predictive given the rows it already holds
        for (size_t pass = 0; pass < rows_each; ++pass) {
            for (size_t g = 0; g < groups; ++g) {
                auto & group = mixture_.groups()[g];
                const Value x = group.sample_value(shared_, rng);
                group.add_value(shared_, x, rng);
                rows_.push_back(x);
                home_.push_back(g);
            }
        }
        mixture_.init(shared_, rng);
    }

    // `updates` row updates, in table order; returns elapsed microseconds
    long churn(size_t updates, rng_t & rng) {
        const long begin = distributions::current_time_us();
        for (size_t u = 0; u < updates; ++u) {
            const size_t i = u % rows_.size();
            if (u % 8 == 0)
                distributions::vector_zero(scores_.size(), scores_.data());
            mixture_.remove_value(shared_, home_[i], rows_[i], rng);
            mixture_.score_value(shared_, rows_[i], scores_, rng);
            mixture_.add_value(shared_, home_[i], rows_[i], rng);
        }
        return distributions::current_time_us() - begin;
    }

    double checksum() const {
        double sum = 0;
        for (float s : scores_) sum += s;
        return sum;
    }

  private:
    typename M::Shared shared_;
    typename M::Mixture mixture_;
    std::vector<Value> rows_;
    std::vector<size_t> home_;
    VectorFloat scores_;
};

template <class M>
void report(size_t largest, rng_t & rng) {
    const std::string name =
        distributions::demangle(typeid(typename M::Shared).name());
    std::printf("%s\nGroups\tcells/us\n", name.c_str());
    const typename M::Shared shared = M::Shared::EXAMPLE();
    double checksum = 0;
    for (size_t groups = 1; groups <= largest; groups *= 10) {
        Table<M> table(shared, groups, 4, rng);
        const size_t updates = 8 * (200 / (1 + groups / 100) + 1);
        const long us = table.churn(updates, rng);
        checksum += table.checksum();
        std::printf("%zu\t%9.4f\n", groups,
                    (double)updates / (double)(us > 0 ? us : 1));
    }
    std::printf("checksum %.6f\n", checksum);
}

}  // namespace

int main(int argc, char ** argv) {
    const size_t largest = argc > 1 ? (size_t)std::atoi(argv[1]) : 1000;
    rng_t rng;
    report<distributions::BetaBernoulli>(largest, rng);
    report<distributions::DirichletDiscrete<4>>(largest, rng);
    report<distributions::GammaPoisson>(largest, rng);
    report<distributions::BetaNegativeBinomial>(largest, rng);
    report<distributions::NormalInverseChiSq>(largest, rng);
    return 0;
}
