"""Generator of tests/golden/dbg_models.json.gz and dbg_low_entropy.json.gz: fixtures
held by the REFERENCE ITSELF for the scorers no compiled reference exists for
here (random.hpp includes Eigen, absent from the image).

Runs in the build container only.  The reference's pure-Python "dbg" flavour

    /root/reference/distributions/dbg/models/{dd,bb,gp,nich,dpd,bnb}.py
    /root/reference/distributions/dbg/clustering.py   (LowEntropy)
    /root/reference/distributions/{mixins,util}.py, dbg/{special,random}.py

is Python 2.  It is read WHERE IT LIES, passed through lib2to3 in memory and
executed from a sys.meta_path finder; nothing of its text is written anywhere.
Two aliases make it importable under numpy 2 (numpy.float/int, which the dbg
modules use as dtypes; numpy.core.umath_tests.inner1d, which vendor/stats.py
imports and this path never calls).

What is dumped (inputs and the reference's outputs; no source):

  per model, per scenario (each module's EXAMPLES — dbg's and lp's, as
  distributions/tests/test_model_flavors.py:61-116 pools them — plus seeded
  random add/remove scripts): the shared's raw dict, the script of
  (op, value), and after every step the group's dump(), Group.score_value on a
  value grid and Group.score_data.

  LowEntropy: score_add_value, score_remove_value, score_counts and
  log_partition_function on grids (dbg/clustering.py:148,170,212,239).

The consumers compare at the reference's own tolerance (tests/util.py:42
TOL = 1e-3, assert_close): tests/test_dbg_goldens.py (oracle, CPU) and
tests/test_gpu_dbg_goldens.py (HIP, through the C ABI).

    python tests/golden/make_dbg_goldens.py
"""
import importlib.abc
import importlib.util
import json
import os
import sys
import types
import warnings

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class _Py2Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Imports distributions.* from the reference tree through lib2to3."""

    def __init__(self):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from lib2to3 import refactor
        fixes = refactor.get_fixers_from_package("lib2to3.fixes")
        self.tool = refactor.RefactoringTool(fixes)

    def _path(self, name):
        rel = name.replace(".", "/")
        for cand in (rel + ".py", rel + "/__init__.py"):
            p = os.path.join(REF, cand)
            if os.path.exists(p):
                return p
        return None

    def find_spec(self, name, path=None, target=None):
        if name != "distributions" and not name.startswith("distributions."):
            return None
        p = self._path(name)
        if p is None:
            return None
        return importlib.util.spec_from_loader(
            name, self, origin=p, is_package=p.endswith("__init__.py"))

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        p = module.__spec__.origin
        if p.endswith("__init__.py"):
            module.__path__ = [os.path.dirname(p)]
        with open(p) as f:
            src = f.read()
        if not src.endswith("\n"):
            src += "\n"
        py3 = str(self.tool.refactor_string(src, p))
        module.__file__ = p
        exec(compile(py3, p, "exec"), module.__dict__)


def import_dbg():
    import numpy
    if not hasattr(numpy, "float"):
        numpy.float = float
    if not hasattr(numpy, "int"):
        numpy.int = int
    shim = types.ModuleType("numpy.core.umath_tests")
    shim.inner1d = lambda a, b: (numpy.asarray(a) * numpy.asarray(b)).sum(-1)
    sys.modules.setdefault("numpy.core.umath_tests", shim)
    sys.meta_path.insert(0, _Py2Finder())
    import distributions.dbg.clustering
    mods = {}
    for name in ("dd", "bb", "gp", "nich", "dpd", "bnb"):
        mods[name] = importlib.import_module("distributions.dbg.models." + name)
    return mods, distributions.dbg.clustering


# The lp modules' EXAMPLES (data; lp/models/dd.pyx:35-48, bb.pyx:35-44,
# gp.pyx:35-40, nich.pyx:35-40, dpd.pyx:37-66, bnb.pyx) are Cython and cannot
# be imported; test_model_flavors pools them with dbg's.  They are listed in
# distributions_amd/lp/models/*.py already (the mirror's own EXAMPLES).
def lp_examples(name):
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    try:
        mod = importlib.import_module("distributions_amd.lp.models." + name)
        return list(mod.EXAMPLES)
    except Exception as e:      # no GPU library here: fall back to dbg's only
        print("  (lp EXAMPLES of %s unavailable: %s)" % (name, e))
        return []


def value_grid(name, shared_raw, values):
    if name == "dd":
        dim = len(shared_raw["alphas"])
        if dim <= 32:
            return list(range(dim))
        return sorted(set(list(values)[:12] + [0, 1, dim // 2, dim - 1]))
    if name == "bb":
        return [False, True]
    if name in ("gp", "bnb"):
        return sorted(set([0, 1, 2, 3, 5, 8, 13, 21, 40, 63, 64, 65, 100, 170]
                          + [int(v) for v in values]))
    if name == "nich":
        return sorted(set([-100.0, -10.0, -3.0, -1.0, -0.25, 0.0, 0.1, 0.5,
                           1.0, 2.5, 7.0, 30.0, 1000.0]
                          + [float(v) for v in values]))
    if name == "dpd":
        return None     # the shared's values, decided after realize()
    raise ValueError(name)


def jsonable(x):
    import numpy
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, numpy.ndarray):
        return x.tolist()
    if isinstance(x, (numpy.integer,)):
        return int(x)
    if isinstance(x, (numpy.floating,)):
        return float(x)
    return x


def random_scripts(name, rs):
    """Seeded shared + add/remove scripts beyond EXAMPLES (BASELINE-like
    parameter ranges: dim up to 256, large counts, extreme reals)."""
    out = []
    if name == "dd":
        for dim in (2, 16, 256):
            alphas = rs.gamma(2.0, 0.5, dim).tolist()
            vals = rs.randint(0, dim, 150 if dim == 256 else 60).tolist()
            out.append(({"alphas": alphas}, vals))
        out.append(({"alphas": [0.5] * 256},
                    (rs.zipf(1.3, 200) % 256).tolist()))
    elif name == "bb":
        for a, b in ((0.5, 2.0), (10.5, 0.5), (1e-2, 30.0)):
            out.append(({"alpha": a, "beta": b},
                        [bool(v) for v in rs.rand(40) < 0.3]))
    elif name == "gp":
        for a, ib, lam in ((1.0, 1.0, 5.0), (3.0, 0.25, 40.0),
                           (0.1, 10.0, 0.3), (50.0, 2.0, 150.0)):
            out.append(({"alpha": a, "inv_beta": ib},
                        rs.poisson(lam, 50).tolist()))
    elif name == "bnb":
        for a, b, r in ((1.0, 1.0, 1), (2.5, 0.75, 3), (20.0, 6.0, 10)):
            out.append(({"alpha": a, "beta": b, "r": r},
                        rs.poisson(4.0, 50).tolist()))
    elif name == "nich":
        for mu, kappa, s2, nu, loc, sc in ((0.0, 1.0, 1.0, 1.0, 0.0, 1.0),
                                           (1.5, 0.3, 2.5, 4.0, 3.0, 10.0),
                                           (-20.0, 5.0, 0.01, 30.0, -20.0, .1),
                                           (0.0, 0.1, 100.0, 0.5, 0.0, 1e3)):
            out.append(({"mu": mu, "kappa": kappa, "sigmasq": s2, "nu": nu},
                        (loc + sc * rs.randn(50)).tolist()))
    elif name == "dpd":
        for gamma, alpha, nv in ((0.5, 0.5, 6), (2.0, 2.0, 30), (5.0, 0.1, 90)):
            out.append(({"gamma": gamma, "alpha": alpha, "betas": {},
                         "counts": {}}, rs.randint(0, nv, 80).tolist()))
    return out


def run_scenario(mod, name, raw_shared, values, rs, removes=True,
                 realize=True):
    """test_model_flavors._test_group's preparation (the shared sees every
    value, then realize()), then a scripted add / remove walk."""
    temp = mod.Shared.from_dict(raw_shared)
    if realize:
        for v in values:
            temp.add_value(v)
        temp.realize()
    raw = jsonable(temp.dump())
    shared = mod.Shared.from_dict(temp.dump())
    grid = value_grid(name, raw_shared, values)
    if grid is None:
        keys = sorted(int(k) for k in shared.betas)
        grid = keys[:24] + keys[-4:] if len(keys) > 28 else keys
        if shared.beta0 > 0:
            grid = grid + [0xFFFFFFFF]
    script = [("add", v) for v in values]
    if removes:
        present = list(values)
        extra = []
        for _ in range(len(values)):
            if present and rs.rand() < 0.6:
                v = present.pop(rs.randint(len(present)))
                extra.append(("remove", v))
            else:
                v = values[rs.randint(len(values))]
                present.append(v)
                extra.append(("add", v))
        script += extra
    group = mod.Group.from_values(shared)
    steps = []

    big = name == "dd" and len(raw_shared["alphas"]) > 32

    def snapshot(op, v):
        step = {
            "op": op, "value": jsonable(v),
            "score_value": [float(group.score_value(shared, g)) for g in grid],
            "score_data": float(group.score_data(shared)),
        }
        if not big or len(steps) % 25 == 0:     # wide dumps: every 25th step
            step["group"] = jsonable(group.dump())
        steps.append(step)

    snapshot("init", None)
    for op, v in script:
        if op == "add":
            group.add_value(shared, v)
        else:
            group.remove_value(shared, v)
        snapshot(op, v)
    out = {"shared": raw, "grid": jsonable(grid), "steps": steps}
    if name == "dpd":
        out["beta0"] = float(shared.beta0)
    return out


def low_entropy(clustering):
    out = []
    for dataset_size in (5, 10, 100, 1000, 10 ** 5, 10 ** 7):
        le = clustering.LowEntropy(dataset_size)
        sizes = sorted(set(s for s in (0, 1, 2, 3, 4, 7, 46, 47, 48, 49, 99,
                                       500, 9999, 10 ** 5 - 1, 10 ** 7 - 1)
                           if s < dataset_size))
        add = []
        for sample_size in sizes:
            for group_size in (0, 1, 2, 3, 10, 100, 9999, 10000, 10001,
                               10 ** 6):
                if group_size > sample_size:
                    continue
                for empties in (1, 3, 10):
                    add.append([group_size, 1, sample_size, empties,
                                le.score_add_value(group_size, 1, sample_size,
                                                   empties)])
        rem = []
        for sample_size in sizes:
            if sample_size == 0:
                continue
            for group_size in (1, 2, 3, 11, 101, 10001, 10002):
                if group_size > sample_size:
                    continue
                rem.append([group_size, 1, sample_size, 1,
                            le.score_remove_value(group_size, 1, sample_size,
                                                  1)])
        counts_cases = [[1], [1, 1], [2, 1], [3, 1, 0], [5], [2, 2, 1],
                        [20, 10, 10, 5, 1, 1], [30] * 3 + [1] * 9,
                        [1000, 500, 250, 1], [9999, 10001, 47, 48],
                        [10 ** 5 - 100, 50, 50], [10 ** 6] * 9 + [999999, 1]]
        cnt = []
        for c in counts_cases:
            if sum(c) <= dataset_size:
                cnt.append([c, le.score_counts(c)])
        lpf = [[n, le.log_partition_function(n)]
               for n in (0, 1, 2, 5, 46, 47, 48, 49, 100, 1000, 10 ** 5,
                         10 ** 7)]
        out.append({"dataset_size": dataset_size, "score_add_value": add,
                    "score_remove_value": rem, "score_counts": cnt,
                    "log_partition_function": lpf})
    return out


def dump_gz(name, obj):
    import gzip
    raw = json.dumps(obj, separators=(",", ":")).encode()
    with open(os.path.join(HERE, name), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as z:
            z.write(raw)


def main():
    import numpy
    mods, clustering = import_dbg()
    models = {}
    for name, mod in mods.items():
        rs = numpy.random.RandomState(20240601 + len(name) * 7
                                      + sum(map(ord, name)))
        numpy.random.seed(4321)      # dpd's stick-breaking draws
        scen = []
        examples = list(mod.EXAMPLES)
        for e in lp_examples(name):
            if e not in examples:
                examples.append(e)
        for e in examples:
            scen.append(dict(run_scenario(mod, name, e["shared"],
                                          list(e["values"]), rs),
                             source="EXAMPLES"))
        for raw, vals in random_scripts(name, rs):
            scen.append(dict(run_scenario(mod, name, raw, vals, rs),
                             source="random"))
        if name == "dpd":       # mass left on unseen values: OTHER scores
            raw = {"gamma": 0.5, "alpha": 1.5,
                   "betas": {0: 0.25, 7: 0.125, 8: 0.25, 300: 0.125},
                   "counts": {0: 1, 7: 2, 8: 4, 300: 1}}
            vals = [0, 7, 0, 8, 300, 7, 0, 8, 8, 300]
            scen.append(dict(run_scenario(mod, name, raw, vals, rs,
                                          realize=False), source="unrealized"))
        models[name] = {"NAME": mod.NAME, "scenarios": scen}
        print("%-5s %-26s %d scenarios, %d steps" % (
            name, mod.NAME, len(scen), sum(len(s["steps"]) for s in scen)))
    dump_gz("dbg_models.json.gz",
            {"generator": "tests/golden/make_dbg_goldens.py",
             "reference": "distributions 2.0.28 dbg flavour",
             "models": models})
    le = low_entropy(clustering)
    dump_gz("dbg_low_entropy.json.gz",
            {"generator": "tests/golden/make_dbg_goldens.py", "cases": le})
    print("low_entropy: %d dataset sizes" % len(le))


if __name__ == "__main__":
    main()
