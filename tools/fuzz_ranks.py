"""Differential fuzz of the multi-rank loop (dist_gibbs_sweep_sharded) with
real peers on one GPU: random world sizes, feature kinds, placements (block /
by value), shard sizes, batch sizes, group counts, priors that found and empty
groups, ranks that look at their engine between passes, short runs -- every
trial held to the oracle's single-process run of the same batch composition
(tests/test_gpu_native_ranks.py's harness).
usage: fuzz_ranks.py [trials] [seed]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pathlib

import test_gpu_native_ranks as h

def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    failures = 0
    t0 = time.time()
    for t in range(trials):
        world = int(rng.choice([2, 2, 3, 4, 8]))
        config = str(rng.choice(["dd", "dd", "bb", "dpd"]))
        placement = str(rng.choice(["block", "skewed"]))
        dim = None
        if config == "dd":
            dim = int(rng.choice([8, 16, 64, 256]))
            if dim >= world and rng.random() < 0.5:
                placement = "value"
        elif config == "dpd":
            dim = int(rng.choice([50, 200, 1000]))
            if rng.random() < 0.5:
                placement = "value"
        k = int(rng.choice([3, 6, 24, 100, 300]))
        n = int(rng.integers(max(8 * k, 600), 12000))
        per = int(rng.integers(60, 1500))
        sweeps = int(rng.integers(2, 6))
        alpha = float(rng.choice([0.5, 1.0, 8.0, 40.0]))
        peekers = tuple(int(r) for r in range(world) if rng.random() < 0.35)
        spec = dict(config=config, N=n, per=per, sweeps=sweeps, K=k, alpha=alpha,
                    peekers=peekers, placement=placement)
        # (one in four on the generic kernels with the host normalising the
        # group set after every sub-sweep: the loop's other branch)
        if rng.random() < 0.25:
            spec["mode"] = 0
        if dim:
            spec["dim"] = dim
        if rng.random() < 0.4:
            batches = -(-(-(-n // world)) // per)
            spec["run_cap"] = int(batches * rng.integers(1, 3))
        with tempfile.TemporaryDirectory() as d:
            p = pathlib.Path(d)
            try:
                fails = h.run(p, world, spec)
                assert fails == [""] * world, fails
                h.check_equal(p, world, spec)
                verdict = "ok"
            except BaseException as e:   # noqa: BLE001  (report and go on)
                failures += 1
                verdict = "FAILED: " + repr(e)[:300]
        print("trial %3d  world %d %-4s %-5s dim %-5s K %3d N %5d per %4d sweeps %d "
              "alpha %-4g peekers %-12s cap %-5s mode %d %s"
              % (t, world, config, placement, dim, k, n, per, sweeps, alpha,
                 peekers, spec.get("run_cap"), spec.get("mode", 2), verdict),
              flush=True)
    print("%d trials, %d failures, %.0f s (seed %d)"
          % (trials, failures, time.time() - t0, seed))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
