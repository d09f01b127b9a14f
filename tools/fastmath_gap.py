"""SURVEY H1, quantified: what the reference's release flags (-O3 -msse4.1
-mfpmath=sse -ffast-math -funsafe-math-optimizations, CMakeLists.txt:17-23)
do to the functions of the oracle that no compiled reference pins
(sample_from_likelihoods, PitmanYor scores, every model's scorer:
oracle/oracle.h).  oracle.c is built twice -- as the tests use it
(-fno-fast-math, -ffp-contract=off: the operation order spelled out in the
source) and under those flags (make -C oracle fast) -- and the reference's
sequential chain (examples/mixture/main.py:236-244) runs under both on the
same rows and seed: the first row whose assignment differs, per model, or
"none".  This is the error bar on "bit-exact against the oracle" for the
unpinned functions.

python tools/fastmath_gap.py [rows] [groups]        (CPU only, ~1 min)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib as ol   # noqa: E402


def shareds(config, dim):
    if config == "dd":
        return [ol.make_shared(ol.DD, alphas=[0.5] * dim)]
    if config == "bb":
        return [ol.make_shared(ol.BB, alpha=0.5, beta=2.0)]
    if config == "gp":
        return [ol.make_shared(ol.GP, alpha=1.0, inv_beta=1.0)]
    if config == "nich":
        return [ol.make_shared(ol.NICH, mu=0.0, kappa=1.0, sigmasq=1.0, nu=1.0)]
    return [ol.make_shared(ol.GP, alpha=1.0, inv_beta=1.0),
            ol.make_shared(ol.NICH, mu=0.0, kappa=1.0, sigmasq=1.0, nu=1.0)]


def values(config, n, dim, rng):
    if config == "dd":
        return [rng.integers(0, dim, n).astype(np.uint32)]
    if config == "bb":
        return [(rng.random(n) < 0.3).astype(np.uint32)]
    if config == "gp":
        return [rng.poisson(5.0, n).astype(np.uint32)]
    if config == "nich":
        return [rng.normal(0, 1, n).astype(np.float32)]
    return [rng.poisson(5.0, n).astype(np.uint32),
            rng.normal(0, 1, n).astype(np.float32)]


def chain(lib, config, n, k, dim, seed):
    rng = np.random.default_rng(20240601)
    vals = values(config, n, dim, rng)
    assign = (np.arange(n) % k).astype(np.uint32)
    m = ol.OracleMixture(1.0, 0.2, shareds(config, dim), lib=lib)
    m.init_from_assignments(vals, assign, k, 1)
    m.gibbs_sequential(0, n, ol.oracle().orc_rng_seed(seed))
    return m.assign.copy(), len(m)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"),
                           "liboracle.so", "fast"])
    fast = ol.bind_oracle(os.path.join(ROOT, "oracle", "_fast",
                                       "liboracle_fast.so"))
    print("sequential chain, N = %d, K = %d + 1, PitmanYor(1, 0.2), seed 12345:"
          " oracle.c as tested vs under the reference's release flags" % (n, k))
    for config, dim in (("dd", 256), ("dd", 16), ("bb", 0), ("gp", 0),
                        ("nich", 0), ("gp_nich", 0)):
        a, ka = chain(None, config, n, k, dim, 12345)
        b, kb = chain(fast, config, n, k, dim, 12345)
        bad = np.nonzero(a != b)[0]
        name = config + ("-%d" % dim if dim else "")
        if bad.size == 0:
            print("  %-8s first divergent row: none (%d rows identical, %d "
                  "groups)" % (name, n, ka))
        else:
            print("  %-8s first divergent row: %d (%d of %d rows differ "
                  "afterwards, %d vs %d groups)"
                  % (name, bad[0], bad.size, n, ka, kb))


if __name__ == "__main__":
    main()
