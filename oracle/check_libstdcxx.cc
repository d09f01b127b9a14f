// TEST INFRASTRUCTURE ONLY.  Pins the oracle's restatement of the entropy
// source against the third-party code the reference actually calls:
// libstdc++'s std::default_random_engine (random_fwd.hpp:34) and
// std::uniform_real_distribution<float> (random.hpp:47-50).
// Prints "<raw> <unif01 bits>" per draw for the seeds given on the command
// line; tests/test_oracle_rng.py compares that with oracle.c.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

// "variates <draws> <alpha> <beta> <seed>...": sample_gamma(rng, alpha) and
// sample_beta_safe(rng, alpha, beta, 1e-6f) exactly as random.hpp:87-119
// spells them (std::gamma_distribution<double> over rng_t, a fresh
// distribution object per draw), with the engine's state after each draw:
// pins dist_sample_gamma / dist_sample_beta_safe, the entropy of
// DirichletProcessDiscrete::Shared::add_value (dpd.hpp:66-74).
static float probe_gamma(std::default_random_engine & rng, float alpha,
                         float beta = 1.f) {
    std::gamma_distribution<double> sampler(alpha, beta);
    return sampler(rng);
}

static int variates(int argc, char ** argv) {
    int draws = atoi(argv[2]);
    float alpha = atof(argv[3]), beta = atof(argv[4]);
    const float min_value = 1e-6f;
    for (int a = 5; a < argc; ++a) {
        unsigned long seed = strtoul(argv[a], nullptr, 10);
        std::default_random_engine g(seed), b(seed);
        printf("seed %lu\n", seed);
        for (int i = 0; i < draws; ++i) {
            float x = probe_gamma(g, alpha, beta);
            float gx = probe_gamma(b, alpha);
            float gy = probe_gamma(b, beta);
            float p;
            if (gx == 0 && gy == 0) {
                std::uniform_real_distribution<float> sampler(0.0, 1.0);
                p = sampler(b) < alpha / (alpha + beta) ? 1.0 : 0.0;
            } else {
                p = gx / (gx + gy);
            }
            float safe = (p + min_value) / (1.f + min_value);
            uint32_t xb, sb;
            memcpy(&xb, &x, 4);
            memcpy(&sb, &safe, 4);
            std::default_random_engine g2 = g, b2 = b;
            printf("%08x %lu %08x %lu\n", xb, (unsigned long)g2(), sb,
                   (unsigned long)b2());
        }
    }
    return 0;
}

int main(int argc, char ** argv) {
    if (argc > 5 && !strcmp(argv[1], "variates")) return variates(argc, argv);
    int draws = argc > 1 ? atoi(argv[1]) : 16;
    for (int a = 2; a < argc; ++a) {
        unsigned long seed = strtoul(argv[a], nullptr, 10);
        std::default_random_engine raw(seed), eng(seed);
        printf("seed %lu\n", seed);
        for (int i = 0; i < draws; ++i) {
            unsigned long x = raw();
            std::uniform_real_distribution<float> sampler(0.0, 1.0);
            float u = sampler(eng);
            uint32_t bits;
            memcpy(&bits, &u, 4);
            printf("%lu %08x\n", x, bits);
        }
    }
    return 0;
}
