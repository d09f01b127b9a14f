// A caller of DirichletProcessDiscrete::Shared written against the
// reference's member names (include/distributions/models/dpd.hpp:59-101):
// add_value / remove_value / realize with an rng_t.  Host side only (no GPU):
// prints the stick so that tests/test_cpp_shim.py can compare it with the
// same calls through the Python mirror.
#include <cstdio>
#include <distributions/models/dpd.hpp>

using namespace distributions;

int main() {
    rng_t rng(7);
    DirichletProcessDiscrete::Shared shared;
    shared.load(2.0f, 2.0f, {}, {}, {});
    const uint32_t values[] = {5, 4, 3, 2, 1, 0, 3, 2, 1};
    for (uint32_t v : values) shared.add_value(v, rng);
    shared.remove_value(5, rng);
    shared.add_value(77, rng);
    DirichletProcessDiscrete::Shared copy = shared;   // deep copy
    shared.realize(rng);
    printf("%zu %.9g %u %d\n", shared.size(), shared.beta0(), rng.state,
           shared.dim);
    printf("%zu %.9g %u %u\n", copy.size(), copy.beta0(), copy.slot(77),
           copy.slot(4));
    for (int i = 0; i < copy.dim; ++i) printf("%.9g ", copy.betas[i]);
    printf("\n");
    DirichletProcessDiscrete::Shared example =
        DirichletProcessDiscrete::Shared::EXAMPLE();
    printf("%d %.9g %.9g %u\n", example.dim, example.p[0], example.p[1],
           example.slot(42));
    return 0;
}
