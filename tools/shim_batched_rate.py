"""What a caller of the Mixture API pays per (value, group) cell: one value per
call (the reference's loop, benchmarks/mixture.cc:104-115) against
Mixture.score_values over a batch.  usage: shim_batched_rate.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distributions_amd.lp.models import dd  # noqa: E402


def main():
    dim = 256
    shared = dd.Shared.from_dict({'alphas': [0.5] * dim})
    rng = np.random.default_rng(0)
    for K in (10, 100, 1000, 10000):
        mixture = dd.Mixture()
        for g in range(K):
            mixture.append(dd.Group.from_values(
                shared, [int(v) for v in rng.integers(0, dim, 4)]))
        mixture.init(shared)
        one = np.zeros(K, np.float32)
        t0 = time.perf_counter()
        reps = 200
        for i in range(reps):
            mixture.score_value(shared, i % dim, one)
        per_value = reps * K / (time.perf_counter() - t0) / 1e6
        line = "K=%-6d per value %8.3f cells/us" % (K, per_value)
        for n in (256, 4096, 65536):
            if n * K > 2 ** 28:
                continue
            values = [int(v) for v in rng.integers(0, dim, n)]
            words = np.array(values, np.uint32)
            acc = np.zeros((n, K), np.float32)
            core = mixture._handle(shared)
            core.score_values(words, acc)
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                core.score_values(words, acc)
            rate = reps * n * K / (time.perf_counter() - t0) / 1e6
            line += " | batch %5d: %9.1f" % (n, rate)
        print(line, flush=True)


if __name__ == "__main__":
    main()
