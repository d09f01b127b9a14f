"""Differential fuzzing of the per-row API (distributions_amd.lp: Shared /
Group / Mixture, PitmanYor.Mixture) against the oracle: random
hyper-parameters and random sequences of add_value / remove_value / add_group /
remove_group / score_value / score_value_group / score_data.
usage: fuzz_lp.py [trials] [first_seed]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402


def make(rng):
    """-> (lp module, lp Shared, oracle Shared, value sampler)"""
    from distributions_amd.lp.models import bb, bnb, dd, dpd, gp, nich
    kind = rng.choice(["dd", "bb", "gp", "nich", "bnb", "dpd"])
    if kind == "dd":
        dim = int(rng.choice([1, 3, 16, 256]))
        alphas = [float(np.float32(a)) for a in rng.uniform(0.05, 3, dim)]
        return (dd, dd.Shared.from_dict({'alphas': alphas}),
                ol.make_shared(ol.DD, alphas=alphas),
                lambda: int(rng.integers(0, dim)))
    if kind == "bb":
        a, b = float(rng.uniform(0.1, 3)), float(rng.uniform(0.1, 3))
        return (bb, bb.Shared.from_dict({'alpha': a, 'beta': b}),
                ol.make_shared(ol.BB, alpha=a, beta=b),
                lambda: bool(rng.random() < 0.4))
    if kind == "gp":
        a, ib = float(rng.uniform(0.2, 4)), float(rng.uniform(0.2, 4))
        return (gp, gp.Shared.from_dict({'alpha': a, 'inv_beta': ib}),
                ol.make_shared(ol.GP, alpha=a, inv_beta=ib),
                lambda: int(rng.poisson(6.0)) if rng.random() < 0.95
                else int(rng.integers(64, 5000)))
    if kind == "bnb":
        a, b = float(rng.uniform(0.2, 3)), float(rng.uniform(0.2, 3))
        r = int(rng.integers(1, 5))
        return (bnb, bnb.Shared.from_dict({'alpha': a, 'beta': b, 'r': r}),
                ol.make_shared(ol.BNB, alpha=a, beta=b, r=r),
                lambda: int(rng.negative_binomial(r, 0.3)))
    if kind == "dpd":
        dim = int(rng.choice([2, 9, 60]))
        betas = (rng.dirichlet(np.ones(dim)) * 0.9).astype(np.float32)
        keys = sorted(int(v) for v in rng.choice(10 * dim, dim, replace=False))
        raw = {'gamma': 0.5, 'alpha': 0.7,
               'betas': {k: float(b) for k, b in zip(keys, betas)},
               'counts': {k: 1 for k in keys}}
        shared = dpd.Shared.from_dict(raw)
        twin = ol.make_shared(ol.DPD, alpha=shared.params.p[0],
                              beta0=shared.params.p[1],
                              betas=shared.params.betas)
        return (dpd, shared, twin, lambda: keys[int(rng.integers(0, dim))])
    p = [float(rng.normal()), float(rng.uniform(0.1, 3)),
         float(rng.uniform(0.1, 3)), float(rng.uniform(0.02, 5))]
    scale = float(rng.uniform(0.01, 50))
    return (nich, nich.Shared.from_dict(dict(zip(
        ['mu', 'kappa', 'sigmasq', 'nu'], p))),
        ol.make_shared(ol.NICH, mu=p[0], kappa=p[1], sigmasq=p[2], nu=p[3]),
        lambda: float(np.float32(rng.normal() * scale)))


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def trial(seed):
    rng = np.random.default_rng(seed)
    L = ol.oracle()
    module, shared, osh, draw = make(rng)
    word = lambda v: module.Group._word(shared, v)   # noqa: E731
    what = "seed %d %s" % (seed, module.NAME)
    mixture = module.Mixture()
    orc = ol.OracleMixture(1.0, 0.0, [osh])
    n_groups = int(rng.integers(1, 12))
    members = []
    for g in range(n_groups):
        start = [draw() for _ in range(int(rng.integers(0, 4)))]
        mixture.append(module.Group.from_values(shared, start))
        L.orc_mix_slave_append_empty(orc.h, 0)
        for v in start:
            L.orc_mix_slave_group_add_value(orc.h, 0, g, word(v))
        members.append(list(start))
    mixture.init(shared)
    L.orc_mix_slave_init(orc.h, 0)
    for step in range(60):
        op = rng.choice(["add", "add", "remove", "score", "score_group",
                         "data", "add_group", "remove_group", "get"])
        k = len(members)
        if op == "add" and k:
            g, v = int(rng.integers(0, k)), draw()
            mixture.add_value(shared, g, v)
            L.orc_mix_slave_add_value(orc.h, 0, g, word(v))
            members[g].append(v)
        elif op == "remove" and k:
            g = int(rng.integers(0, k))
            if members[g]:
                v = members[g].pop(int(rng.integers(0, len(members[g]))))
                mixture.remove_value(shared, g, v)
                L.orc_mix_slave_remove_value(orc.h, 0, g, word(v))
        elif op == "score" and k:
            v = draw()
            noise = rng.normal(size=k).astype(np.float32)
            got, want = noise.copy(), noise.copy()
            mixture.score_value(shared, v, got)
            L.orc_mix_slave_score_value(orc.h, 0, word(v), want)
            if not np.array_equal(bits(got), bits(want)):
                return what + " step %d score_value" % step
        elif op == "score_group" and k:
            g, v = int(rng.integers(0, k)), draw()
            got = mixture.score_value_group(shared, g, v)
            want = L.orc_mix_slave_score_value_group(orc.h, 0, g, word(v))
            if bits([got])[0] != bits([want])[0]:
                return what + " step %d score_value_group" % step
        elif op == "data":
            got = mixture.score_data(shared)
            want = L.orc_mix_slave_score_data(orc.h, 0)
            exact = module.NAME != 'DirichletProcessDiscrete'
            if (bits([got])[0] != bits([want])[0] if exact
                    else abs(got - want) > 1e-5 * (1 + abs(want))):
                return what + " step %d score_data %r %r" % (step, got, want)
        elif op == "add_group":
            mixture.add_group(shared)
            L.orc_mix_slave_add_group(orc.h, 0)
            members.append([])
        elif op == "remove_group" and k > 1:
            g = int(rng.integers(0, k))
            mixture.remove_group(shared, g)
            L.orc_mix_slave_remove_group(orc.h, 0, g)
            members[g] = members[-1]
            members.pop()
        elif op == "get" and k:
            g = int(rng.integers(0, k))
            if not np.array_equal(mixture[g].words, orc.get_group(0, g)):
                return what + " step %d group %d statistics" % (step, g)
        if len(mixture) != L.orc_mix_slave_size(orc.h, 0):
            return what + " step %d size" % step
    return None


def trial_growing_dpd(seed):
    """A DirichletProcessDiscrete feature whose Shared starts (nearly) empty:
    values appear through Shared.add_value (the stick breaks, dpd.hpp:66-74),
    vanish through remove_value (:76-83) and their dense slots are taken
    again, while a device mixture lives on that Shared.  Every few steps the
    mixture's scores over all live values and OTHER, score_values, and the
    groups' counts are compared with an oracle mixture built from scratch on
    the Shared's current dense view with the same memberships (bit for bit)."""
    from distributions_amd.lp import random as lprandom
    from distributions_amd.lp.models import dpd
    rng = np.random.default_rng(seed)
    L = ol.oracle()
    lprandom.seed(int(rng.integers(1, 2 ** 31 - 1)))
    what = "seed %d growing DPD" % seed
    start = {int(v): 1 for v in rng.choice(50, int(rng.integers(0, 4)),
                                           replace=False)}
    betas = {v: float(b) for v, b in zip(
        start, rng.dirichlet(np.ones(len(start) + 1))[:len(start)] * 0.6)}
    shared = dpd.Shared.from_dict({'gamma': float(rng.uniform(0.3, 6)),
                                   'alpha': float(rng.uniform(0.2, 3)),
                                   'betas': betas, 'counts': start})
    K = int(rng.integers(1, 7))
    members = [[] for _ in range(K)]
    mixture = dpd.Mixture()
    for _ in range(K):
        mixture.append(dpd.Group.from_values(shared))
    mixture.init(shared)
    universe = int(rng.choice([6, 40]))

    def compare(step):
        p = shared.params
        if p.dim == 0:
            return None
        osh = ol.make_shared(ol.DPD, alpha=p.p[0], beta0=p.p[1], betas=p.betas)
        orc = ol.OracleMixture(1.0, 0.0, [osh])
        for g in range(K):
            L.orc_mix_slave_append_empty(orc.h, 0)
            for v in members[g]:
                L.orc_mix_slave_group_add_value(orc.h, 0, g, shared.remap(v))
        L.orc_mix_slave_init(orc.h, 0)
        live = [v for v in shared.values if v is not None] + [dpd.OTHER]
        batch = np.zeros((len(live), K), np.float32)
        mixture.score_values(shared, live, batch)
        for i, v in enumerate(live):
            got = np.zeros(K, np.float32)
            want = np.zeros(K, np.float32)
            mixture.score_value(shared, v, got)
            L.orc_mix_slave_score_value(orc.h, 0, shared.remap(v), want)
            if not np.array_equal(bits(got), bits(want)):
                return what + " step %d score_value(%r)" % (step, v)
            if not np.array_equal(bits(batch[i]), bits(want)):
                return what + " step %d score_values(%r)" % (step, v)
        for g in range(K):
            want = {}
            for v in members[g]:
                want[v] = want.get(v, 0) + 1
            if mixture[g].dump()['counts'] != want:
                return what + " step %d counts of group %d" % (step, g)
        mixture.validate(shared)
        return None

    for step in range(80):
        op = rng.choice(["add", "add", "add", "remove", "remove", "check"])
        if op == "add":
            v = int(rng.integers(0, universe))
            if shared.beta0 <= 0 and v not in shared.dump()['betas']:
                continue   # (the stick is used up: dpd.hpp:69)
            shared.add_value(v)
            g = int(rng.integers(0, K))
            mixture.add_value(shared, g, v)
            members[g].append(v)
        elif op == "remove":
            g = int(rng.integers(0, K))
            if members[g]:
                v = members[g].pop(int(rng.integers(0, len(members[g]))))
                mixture.remove_value(shared, g, v)
                # (the values the Shared was loaded with keep the one row
                # they came with; every other value vanishes with its last)
                shared.remove_value(v)
        else:
            err = compare(step)
            if err:
                return err
    return compare(80)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    failures = 0
    for seed in range(first, first + trials):
        try:
            err = trial_growing_dpd(seed) if seed % 5 == 4 else trial(seed)
        except Exception as e:   # noqa: BLE001
            err = "seed %d: exception %r" % (seed, e)
        if err:
            failures += 1
            print("FAIL", err, flush=True)
    print("%d trials, %d failures" % (trials, failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
