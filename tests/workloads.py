"""Seeded synthetic workloads shared by the parity tests and bench.py
(SURVEY 8d): identical bytes for the oracle and the GPU."""
import numpy as np

import oracle_lib as ol
from distributions_amd import engine

SEED = 20240601


def make(config, n, k, seed=SEED, dim=None):
    """-> (oracle shareds, engine shareds, values per feature, assign_packed)"""
    rng = np.random.default_rng(seed)
    assign = (np.arange(n) % k).astype(np.uint32)
    if config == "dd":
        dim = dim or 16
        vals = [rng.integers(0, dim, n).astype(np.uint32)]
        osh = [ol.make_shared(ol.DD, alphas=[0.5] * dim)]
        gsh = [engine.dd_shared([0.5] * dim)]
    elif config == "dd_zipf":
        # SURVEY 8d's skewed variant of C2: Zipf(s = 1.1) values, the same
        # symmetric prior
        dim = dim or 256
        p = 1.0 / np.arange(1, dim + 1) ** 1.1
        vals = [rng.choice(dim, n, p=p / p.sum()).astype(np.uint32)]
        osh = [ol.make_shared(ol.DD, alphas=[0.5] * dim)]
        gsh = [engine.dd_shared([0.5] * dim)]
    elif config == "dd_skew":
        dim = dim or 16
        p = 1.0 / np.arange(1, dim + 1) ** 1.1
        vals = [rng.choice(dim, n, p=p / p.sum()).astype(np.uint32)]
        alphas = [2.0 / (i + 1) for i in range(dim)]
        osh = [ol.make_shared(ol.DD, alphas=alphas)]
        gsh = [engine.dd_shared(alphas)]
    elif config == "bb":
        vals = [(rng.random(n) < 0.3).astype(np.uint32)]
        osh = [ol.make_shared(ol.BB, alpha=0.5, beta=2.0)]
        gsh = [engine.bb_shared(0.5, 2.0)]
    elif config == "gp":
        vals = [rng.poisson(5.0, n).astype(np.uint32)]
        osh = [ol.make_shared(ol.GP, alpha=1.0, inv_beta=1.0)]
        gsh = [engine.gp_shared(1.0, 1.0)]
    elif config == "bnb":
        vals = [rng.negative_binomial(3, 0.4, n).astype(np.uint32)]
        osh = [ol.make_shared(ol.BNB, alpha=1.5, beta=0.75, r=3)]
        gsh = [engine.bnb_shared(1.5, 0.75, 3)]
    elif config == "nich":
        vals = [rng.normal(0, 1, n).astype(np.float32)]
        osh = [ol.make_shared(ol.NICH, mu=0.0, kappa=1.0, sigmasq=1.0, nu=1.0)]
        gsh = [engine.nich_shared(0.0, 1.0, 1.0, 1.0)]
    elif config == "gp_nich":
        vals = [rng.poisson(5.0, n).astype(np.uint32),
                rng.normal(0, 1, n).astype(np.float32)]
        osh = [ol.make_shared(ol.GP, alpha=1.0, inv_beta=1.0),
               ol.make_shared(ol.NICH, mu=0.0, kappa=1.0, sigmasq=1.0, nu=1.0)]
        gsh = [engine.gp_shared(1.0, 1.0),
               engine.nich_shared(0.0, 1.0, 1.0, 1.0)]
    elif config == "nich2":
        vals = [rng.normal(0, 1, n).astype(np.float32),
                rng.normal(3, 2, n).astype(np.float32)]
        osh = [ol.make_shared(ol.NICH, mu=0.0, kappa=1.0, sigmasq=1.0, nu=1.0),
               ol.make_shared(ol.NICH, mu=1.0, kappa=0.5, sigmasq=2.0, nu=3.0)]
        gsh = [engine.nich_shared(0.0, 1.0, 1.0, 1.0),
               engine.nich_shared(1.0, 0.5, 2.0, 3.0)]
    elif config == "dpd":
        dim = dim or 100
        betas = np.full(dim, 1.0 / dim, np.float32)
        vals = [rng.integers(0, dim, n).astype(np.uint32)]
        osh = [ol.make_shared(ol.DPD, alpha=0.5, betas=betas, beta0=0.0)]
        gsh = [engine.dpd_shared(0.5, betas, 0.0)]
    elif config == "dpd_other":
        dim = dim or 50
        betas = np.full(dim, 0.9 / dim, np.float32)
        v = rng.integers(0, dim, n).astype(np.uint32)
        # dpd.hpp:56: OTHER scores with alpha * beta0 and is never added or
        # removed (dpd.hpp:193,212); the engine treats it as a scored-only
        # value, so keep it out of the statistics by never using it in rows
        vals = [v]
        osh = [ol.make_shared(ol.DPD, alpha=0.5, betas=betas, beta0=0.1)]
        gsh = [engine.dpd_shared(0.5, betas, 0.1)]
    elif config == "dd_bb_gp":
        dim = dim or 8
        vals = [rng.integers(0, dim, n).astype(np.uint32),
                (rng.random(n) < 0.5).astype(np.uint32),
                rng.poisson(3.0, n).astype(np.uint32)]
        osh = [ol.make_shared(ol.DD, alphas=[0.5] * dim),
               ol.make_shared(ol.BB, alpha=0.5, beta=2.0),
               ol.make_shared(ol.GP, alpha=1.0, inv_beta=1.0)]
        gsh = [engine.dd_shared([0.5] * dim), engine.bb_shared(0.5, 2.0),
               engine.gp_shared(1.0, 1.0)]
    else:
        raise ValueError(config)
    return osh, gsh, vals, assign


def planted(n, k_true=64, seed=1, n_cat=4, dim=16, n_real=2):
    """A planted mixture (rows that DO have structure, unlike the bench's
    noise): k_true clusters, each with its own peaked categorical law for
    n_cat DirichletDiscrete(dim) features and its own mean for n_real
    NormalInverseChiSq features.
    -> (truth[n], oracle shareds, engine shareds, values per feature)"""
    rng = np.random.default_rng(seed)
    z = rng.integers(0, k_true, n)
    vals, osh, gsh = [], [], []
    for _ in range(n_cat):
        theta = rng.dirichlet([0.1] * dim, k_true)
        cdf = np.cumsum(theta, 1)
        u = rng.random(n)
        x = (u[:, None] > cdf[z]).sum(1).clip(max=dim - 1).astype(np.uint32)
        vals.append(x)
        osh.append(ol.make_shared(ol.DD, alphas=[0.5] * dim))
        gsh.append(engine.dd_shared([0.5] * dim))
    for _ in range(n_real):
        mu = rng.normal(0, 4, k_true)
        x = (mu[z] + rng.normal(0, 0.5, n)).astype(np.float32)
        vals.append(x)
        osh.append(ol.make_shared(ol.NICH, mu=0.0, kappa=0.1, sigmasq=1.0,
                                  nu=1.0))
        gsh.append(engine.nich_shared(0.0, 0.1, 1.0, 1.0))
    return z, osh, gsh, vals


def adjusted_rand_index(a, b):
    ua, ia = np.unique(a, return_inverse=True)
    ub, ib = np.unique(b, return_inverse=True)
    c = np.zeros((ua.size, ub.size), np.int64)
    np.add.at(c, (ia, ib), 1)

    def pairs(x):
        return x * (x - 1) / 2.0
    s, sa, sb = pairs(c).sum(), pairs(c.sum(1)).sum(), pairs(c.sum(0)).sum()
    e = sa * sb / pairs(a.size)
    return float((s - e) / (0.5 * (sa + sb) - e))
