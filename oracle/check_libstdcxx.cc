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

int main(int argc, char ** argv) {
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
