// One Gibbs row update written against include/distributions_hip.hpp exactly
// as a downstream C++ program writes it against the reference's headers
// (the loop of benchmarks/mixture.cc:104-115 plus the driver and
// the sampler).  Build:
//   g++ -std=c++11 -Iinclude examples/row_update.cc -o examples/row_update
//       -Ldistributions_amd -ldistributions_hip -Wl,-rpath,$PWD/distributions_amd
#include <cstdio>

#include "distributions_hip.hpp"

using namespace distributions_hip;

int main() {
    typedef DirichletDiscrete<> Model;
    rng_t rng(1);

    Model::Shared shared;
    shared.dim = 4;
    for (int i = 0; i < 4; ++i) shared.alphas[i] = 0.5f;
    PitmanYor py = {1.0f, 0.2f};

    const int values[8] = {0, 1, 0, 2, 0, 1, 0, 3};
    size_t assignments[8];
    Model::Mixture slave;
    PitmanYor::Mixture driver;
    MixtureIdTracker tracker;
    const int K = 3;
    for (int k = 0; k < K + 1; ++k) {          // K groups + 1 empty
        Model::Group group;
        group.init(shared, rng);
        for (int i = k; k < K && i < 8; i += K) group.add_value(shared, values[i], rng);
        slave.append(shared, group);
        driver.counts().push_back(k < K ? (8 - k + K - 1) / K : 0);
    }
    for (int i = 0; i < 8; ++i) assignments[i] = i % K;
    slave.init(shared, rng);
    driver.init(py);
    tracker.init(K + 1);

    for (int pass = 0; pass < 3; ++pass) {
        for (int i = 0; i < 8; ++i) {
            size_t g = tracker.global_to_packed((uint32_t)assignments[i]);
            bool removed = driver.remove_value(py, g);
            slave.remove_value(shared, g, values[i], rng);
            if (removed) {
                slave.remove_group(shared, g);
                tracker.remove_group((uint32_t)g);
            }
            VectorFloat scores(driver.size());
            driver.score_value(py, scores);                    // overwrite
            slave.score_value(shared, values[i], scores, rng); // accumulate
            size_t g2 = sample_from_scores_overwrite(rng, scores);
            bool added = driver.add_value(py, g2);
            slave.add_value(shared, g2, values[i], rng);
            if (added) {
                slave.add_group(shared, rng);
                tracker.add_group();
            }
            assignments[i] = tracker.packed_to_global((uint32_t)g2);
        }
    }
    printf("groups %zu assignments", driver.size());
    for (int i = 0; i < 8; ++i) printf(" %zu", assignments[i]);
    printf("\n");

    // either side of the update: a hyper-parameter grid in one call
    // (mixture.hpp:433-438) and a group checkpointed in the reference's wire
    // format (dd.hpp:94-111)
    std::vector<Model::Shared> grid(3, shared);
    grid[1].alphas[2] = 1.5f;
    grid[2].alphas[2] = 1.5f;
    grid[2].alphas[0] = 0.25f;
    VectorFloat grid_scores(grid.size());
    slave.score_data_grid(grid, grid_scores, rng);
    printf("grid %.6f %.6f %.6f single %.6f\n", grid_scores[0], grid_scores[1],
           grid_scores[2], slave.score_data(shared, rng));
    Model::Group first = slave.groups(shared, 0);
    const std::string wire = first.protobuf_dump(shared);
    Model::Group back;
    back.protobuf_load(shared, wire);
    printf("wire %zu bytes roundtrip %s\n", wire.size(),
           back.words == first.words ? "ok" : "MISMATCH");
    return 0;
}
