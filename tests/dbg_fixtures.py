"""Loader of the fixtures the reference's own Python ("dbg") flavour produced
(tests/golden/make_dbg_goldens.py) and the comparison the reference's tests
use between flavours (distributions/tests/util.py:42,100-140: TOL = 1e-3,
|a - b| < TOL * (1 + |a| + |b|)).  Shared by the oracle-side test (CPU) and
the HIP-side test (GPU)."""
import gzip
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-3      # distributions/tests/util.py:42
OTHER = 0xFFFFFFFF


def _load(name):
    with gzip.open(os.path.join(HERE, "golden", name), "rt") as f:
        return json.load(f)


_CACHE = {}


def models():
    if "m" not in _CACHE:
        _CACHE["m"] = _load("dbg_models.json.gz")["models"]
    return _CACHE["m"]


def low_entropy():
    if "le" not in _CACHE:
        _CACHE["le"] = _load("dbg_low_entropy.json.gz")["cases"]
    return _CACHE["le"]


def scenario_ids():
    return [(name, i) for name in ("dd", "bb", "gp", "nich", "dpd", "bnb")
            for i in range(len(models()[name]["scenarios"]))]


def assert_close(got, want, msg="", allow=0.0):
    """util.assert_close for floats (distributions/tests/util.py:110-120);
    `allow` is an absolute allowance on top, zero unless the caller can name
    the table of the reference's C++ flavour that produces it"""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    diff = np.abs(got - want)
    norm = 1.0 + np.abs(got) + np.abs(want)
    bad = ~(diff < TOL * norm + allow)
    assert not bad.any(), "%s: got %r, dbg flavour %r (off by %g)" % (
        msg, got[bad][:4], want[bad][:4], float((diff / norm)[bad].max()))


def dpd_dense(shared_raw):
    """DirichletProcessDiscrete's values -> 0..V-1 in sorted order (the
    dense remap distributions_amd.lp.models.dpd uses) and the dense betas"""
    keys = sorted(int(k) for k in shared_raw["betas"])
    index = {k: i for i, k in enumerate(keys)}
    betas = [float(shared_raw["betas"][str(k)]) for k in keys]
    return keys, index, betas
