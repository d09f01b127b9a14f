"""Generator of tests/golden/sampler_probs.json.gz: what the REFERENCE says the
softmax of a score vector is -- distributions/util.py:33-38 scores_to_probs,
the function its own Python flavour samples with (dbg/random.py:63-65
sample_discrete_log) and its tests hold the C++ sampler to
(tests/test_random.py:183-247) -- on score vectors of the sizes the Gibbs row
update meets (SURVEY 8(c) golden set (3)).

Runs in the build container only; the reference's Python 2 modules are read
where they lie through the lib2to3 finder of make_dbg_goldens.py (nothing of
their text is written anywhere).  Inputs and outputs only:

    python tests/golden/make_sampler_goldens.py
"""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_dbg_goldens as gen   # noqa: E402  (the finder)


def main():
    gen.import_dbg()
    from distributions.util import scores_to_probs
    rng = np.random.default_rng(20240601)
    cases = [{"name": "survey probe (SURVEY 8c golden set 3)",
              "scores": [-1.0, -2.5, 0.25, -0.75, -3.0]}]
    for k in (1, 2, 5, 64, 1024):
        for spread in (0.5, 4.0):
            s = rng.normal(-3.0, spread, k).astype(np.float32)
            cases.append({"name": "normal(-3, %g), K = %d" % (spread, k),
                          "scores": [float(v) for v in s]})
    # the shape of a Gibbs row's scores: a few groups carry the mass
    s = rng.normal(-9.0, 1.0, 1024).astype(np.float32)
    s[[3, 500, 1023]] = [-1.0, -1.5, -2.0]
    cases.append({"name": "three heavy groups among 1024",
                  "scores": [float(v) for v in s]})
    for c in cases:
        f32 = np.asarray(c["scores"], np.float32)
        c["probs"] = [float(p) for p in scores_to_probs(f32.astype(np.float64))]
    path = os.path.join(HERE, "sampler_probs.json.gz")
    with gzip.open(path, "wt") as f:
        json.dump({"source": "distributions/util.py:33-38 scores_to_probs",
                   "cases": cases}, f)
    print("wrote", path, len(cases), "cases")


if __name__ == "__main__":
    main()
