"""The table behind DESIGN.md's "how large may a frozen sub-sweep be": the
sequential chain against batch chains on a planted mixture (tests/
test_batch_validity.py has the assertions).  Runs the CPU oracle; with --gpu
the engine itself (bit-identical, faster).
usage: python tools/batch_validity.py [--rows 200000] [--sweeps 16] [--gpu]
                                      [--batches 4096,20000,65536,200000]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np   # noqa: E402
import oracle_lib as ol   # noqa: E402
import workloads   # noqa: E402
import test_batch_validity as tbv   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200_000)
    ap.add_argument("--sweeps", type=int, default=16)
    ap.add_argument("--batches", default="4096,20000,65536,200000")
    ap.add_argument("--gpu", action="store_true")
    args = ap.parse_args()
    n, k = args.rows, 64
    truth, osh, gsh, vals = workloads.planted(n, k)
    start = (np.arange(n) % k).astype(np.uint32)
    if args.gpu:
        import test_gpu_batch_validity as g
        run = lambda b: g.run_chain(gsh, vals, start, k, b, args.sweeps)  # noqa: E731
    else:
        run = lambda b: tbv.run_chain(osh, vals, start, k, b, args.sweeps)  # noqa: E731
    print("| B | B/N | groups | ARI | sweeps to 90 % of the gain | "
          "score/row after each sweep |")
    print("|---|---|---|---|---|---|")
    ref = None
    for b in [0] + [int(x) for x in args.batches.split(",") if x]:
        t0 = time.time()
        traj, assign = run(b)
        if ref is None:
            ref = traj
        level = ref[0] + 0.9 * (ref[-1] - ref[0])
        print("| %s | %s | %d | %.3f | %d | %s | (%.1f s)" % (
            b or "sequential", "%.3f" % (b / n) if b else "-",
            np.unique(assign).size,
            workloads.adjusted_rand_index(truth, assign),
            tbv.sweeps_to_reach(traj, level),
            " ".join("%.2f" % x for x in traj), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
