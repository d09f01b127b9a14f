"""The multi-rank driver (distributions_amd.engine.ShardedGibbs) under gloo,
world_size 2, on CPU.  The per-rank compute object is the oracle stand-in
(oracle_lib.OracleBackend): this exercises row sharding, the statistic-delta
all-reduce, lock-step normalisation and the global draw indexing -- not the
HIP kernels (those are covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

N, K, BATCH, SWEEPS, SEED = 2400, 12, 300, 2, 4242
# BASELINE config 4's shape (SURVEY 8d "C4"): PitmanYor(alpha=1, d=0.2), one
# DirichletDiscrete(256) feature, 8 ranks, sub-sweeps of 65 536 rows per rank
# -- at a reduced N (two sub-sweeps per rank and pass)
C4 = dict(N=8 * 131072, K=48, BATCH=8 * 65536, SWEEPS=1, world=8)


def shape(config):
    if config == "c4":
        return C4["N"], C4["K"], C4["BATCH"], C4["SWEEPS"]
    return N, K, BATCH, SWEEPS


def make_rows(config):
    """-> (list of value columns, packed assignment)"""
    n, k, _, _ = shape(config)
    rng = np.random.default_rng(7)
    assign = (np.arange(n) % k).astype(np.uint32)
    if config == "dd":
        return [rng.integers(0, 8, n).astype(np.uint32)], assign
    if config == "c4":
        return [rng.integers(0, 256, n).astype(np.uint32)], assign
    # order-dependent statistics: GammaPoisson (log_prod) + NormalInverseChiSq
    return [rng.poisson(5.0, n).astype(np.uint32),
            rng.normal(0.0, 1.0, n).astype(np.float32)], assign


def make_mix(config, values, assign, lo, hi):
    import oracle_lib as ol
    k = shape(config)[1]
    if config == "dd":
        sh = [ol.make_shared(ol.DD, alphas=[0.5] * 8)]
    elif config == "c4":
        sh = [ol.make_shared(ol.DD, alphas=[0.5] * 256)]
    else:
        sh = [ol.make_shared(ol.GP, alpha=1.0, inv_beta=1.0),
              ol.make_shared(ol.NICH, mu=0.0, kappa=1.0, sigmasq=1.0, nu=1.0)]
    m = (ol.OracleMixture(1.0, 0.2, sh) if config == "c4"
         else ol.OracleMixture(3.0, 0.3, sh))
    m.init_from_assignments([v[lo:hi] for v in values], assign[lo:hi], k, 2)
    return m


def group_words(m):
    return np.stack([np.concatenate([m.get_group(f, g)
                                     for f in range(len(m.shareds))])
                     for g in range(len(m))])


def worker(rank, world, port, out, config):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as ol
    from distributions_amd import engine
    N, K, BATCH, SWEEPS = shape(config)
    values, assign = make_rows(config)
    lo, hi = rank * N // world, (rank + 1) * N // world
    m = make_mix(config, values, assign, lo, hi)
    backend = ol.OracleBackend(m, row_offset=lo)
    # the oracle's value words are what a caller hands to load_rows
    columns = [torch.from_numpy(w.view(np.int32).copy()) for w in m.values]
    packed = torch.from_numpy(assign[lo:hi].view(np.int32).copy())
    sharded = engine.ShardedGibbs(backend, hi - lo, lo, device="cpu",
                                  columns=columns, assign_packed=packed)
    sharded.sync_initial_stats()
    st = ol.oracle().orc_rng_seed(SEED)
    for s in range(SWEEPS):
        sharded.sweep(BATCH // world, st, draw_base=s * N)
    np.save(os.path.join(out, "assign_%d.npy" % rank), m.assign)
    np.save(os.path.join(out, "counts_%d.npy" % rank), m.counts())
    np.save(os.path.join(out, "groups_%d.npy" % rank), group_words(m))
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("config,world", [("dd", 2), ("gp_nich", 2),
                                          ("c4", C4["world"])])
def test_ranks_equal_one_rank_with_the_same_batches(tmp_path, config, world):
    """world_size 2 (both exchanges: integer deltas, ordered rows) and
    world_size 8 in BASELINE config 4's shape."""
    import oracle_lib as ol
    N, K, BATCH, SWEEPS = shape(config)
    mp.spawn(worker, args=(world, free_port(), str(tmp_path), config),
             nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("assign_%d.npy" % r))
                          for r in range(world)])
    counts = [np.load(tmp_path / ("counts_%d.npy" % r)) for r in range(world)]
    groups = [np.load(tmp_path / ("groups_%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(counts[0], counts[r])   # replicas agree
        assert np.array_equal(groups[0], groups[r])   # ... bit for bit

    # single process, same batch composition: batch b = the union over ranks
    # of local rows [b*B/2, (b+1)*B/2) of each shard
    values, assign = make_rows(config)
    m = make_mix(config, values, assign, 0, N)
    L = ol.oracle()
    ol._phase_sigs(L)
    st = L.orc_rng_seed(SEED)
    half = N // world
    per = BATCH // world
    for s in range(SWEEPS):
        for b in range(0, half, per):
            snap = m.counts().copy()
            moves = []
            for r in range(world):
                r0, r1 = r * half + b, r * half + min(half, b + per)
                old = np.zeros(r1 - r0 + 1, np.uint32)
                new = np.zeros(r1 - r0 + 1, np.uint32)
                L.orc_mix_batch_sample(m.h, r0, r1, m._vals, m.assign, st,
                                       s * N, 0, old, new)
                moves.append((r0, r1, old, new))
            for r0, r1, old, new in moves:
                L.orc_mix_apply_moves(m.h, r0, r1, m._vals, m.assign, old, new)
            L.orc_mix_batch_finish(m.h, np.ascontiguousarray(snap, np.int32))
    assert np.array_equal(m.counts(), counts[0])
    assert np.array_equal(got, m.assign)
    # statistics too, the order-dependent floats included
    assert np.array_equal(group_words(m), groups[0])
