"""Soak: many consecutive sweeps of device-normalised runs under group churn,
the state looked at only at the end (and once in the middle), against the
oracle.  usage: soak_open_runs.py [sweeps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
import workloads  # noqa: E402
from distributions_amd import engine  # noqa: E402


def main():
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n, k = 6000, 300
    osh, gsh, vals, assign = workloads.make("dd", n, k, dim=16)
    bad = 0
    for empty, batch in ((1, 1500), (3, 777)):
        orc = ol.OracleMixture(20.0, 0.5, osh)
        orc.init_from_assignments(vals, assign, k, empty)
        gpu = engine.Gibbs(20.0, 0.5, gsh)
        gpu.set_option("value_sorted", 2)
        gpu.load_rows(vals, assign, k, empty)
        seed = 4711
        st = ol.oracle().orc_rng_seed(seed)
        for s in range(sweeps):
            for b in range(0, n, batch):
                orc.gibbs_batch(b, min(n, b + batch), st, s * n)
            gpu.sweep(0, n, batch, seed, draw_base=s * n)
            if s == sweeps // 2 or s == sweeps - 1:
                report = gpu.validate(raise_on_failure=False)
                if report["code"]:
                    print("validate:", report, flush=True)
                    bad += 1
                same = (len(gpu) == len(orc)
                        and np.array_equal(gpu.counts(), orc.counts())
                        and np.array_equal(gpu.assignments(), orc.assign))
                print("empty=%d batch=%d sweep %d: %d groups, %s" % (
                    empty, batch, s, len(gpu), "same" if same else "DIFFERENT"),
                    flush=True)
                bad += not same
        print(gpu.core.debug_counts())
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
