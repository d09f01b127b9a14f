"""The multi-rank driver with the REAL engine on every rank: two processes
share the one GPU of the test box, the collectives go through gloo (staged
through host memory by engine.ShardedGibbs), the kernels are the product's.
Same expectation as tests/test_sharded_gloo.py: the two-rank run equals one
process sampling the same batch composition -- assignments, group sizes and
every statistic, the order-dependent floats included."""
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

pytestmark = pytest.mark.gpu

K, PER_RANK_BATCH, SWEEPS, SEED = 24, 750, 2, 777


def worker(rank, world, port, out, config, mode, N, merged=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as ol
    import workloads
    from distributions_amd import _core, engine
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    osh, gsh, vals, assign = workloads.make(config, N, K)
    lo, hi = rank * N // world, (rank + 1) * N // world
    cols = [torch.from_numpy(ol.value_words(s.kind, v[lo:hi]).view(np.int32)
                             .copy()).to(dev) for s, v in zip(osh, vals)]
    packed = torch.from_numpy(assign[lo:hi].view(np.int32).copy()).to(dev)
    gpu = engine.Gibbs(1.0, 0.2, gsh)
    gpu.set_option("value_sorted", mode)
    gpu.set_option("float_stats", merged)
    gpu.load_rows_torch(cols, packed.clone(), K, 2, row_offset=lo)
    sharded = engine.ShardedGibbs(gpu.core, hi - lo, lo, device=dev,
                                  columns=cols, assign_packed=packed)
    sharded.sync_initial_stats()
    for s in range(SWEEPS):
        sharded.sweep(PER_RANK_BATCH, _core.rng_seed(SEED), draw_base=s * N)
    torch.cuda.synchronize()
    np.save(os.path.join(out, "assign_%d.npy" % rank), gpu.assignments())
    np.save(os.path.join(out, "counts_%d.npy" % rank), gpu.counts())
    np.save(os.path.join(out, "groups_%d.npy" % rank), np.stack([
        np.concatenate([gpu.get_group(f, g) for f in range(len(gsh))])
        for g in range(len(gpu))]))
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("config,mode,world,N", [
    ("dd", 2, 2, 6000), ("dd", 0, 2, 6000), ("gp_nich", 1, 2, 6000),
    ("dd_bb_gp", 1, 2, 6000),
    # three ranks, shards of unequal length (the last batch is ragged and one
    # rank runs out of rows first: padded exchanges)
    ("dd", 2, 3, 5501), ("gp_nich", 1, 3, 5501)])
def test_ranks_on_one_gpu_equal_one_process(tmp_path, config, mode, world, N):
    import oracle_lib as ol
    import workloads
    mp.spawn(worker, args=(world, free_port(), str(tmp_path), config, mode, N),
             nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("assign_%d.npy" % r))
                          for r in range(world)])
    counts = [np.load(tmp_path / ("counts_%d.npy" % r)) for r in range(world)]
    groups = [np.load(tmp_path / ("groups_%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(counts[0], counts[r])   # replicas agree ...
        assert np.array_equal(groups[0], groups[r])   # ... bit for bit

    osh, gsh, vals, assign = workloads.make(config, N, K)
    m = ol.OracleMixture(1.0, 0.2, osh)
    m.init_from_assignments(vals, assign, K, 2)
    L = ol.oracle()
    ol._phase_sigs(L)
    st = L.orc_rng_seed(SEED)
    per = PER_RANK_BATCH
    bounds = [(r * N // world, (r + 1) * N // world) for r in range(world)]
    longest = max(hi - lo for lo, hi in bounds)
    for s in range(SWEEPS):
        for b in range(0, longest, per):
            snap = m.counts().copy()
            moves = []
            for lo, hi in bounds:
                r0, r1 = min(hi, lo + b), min(hi, lo + b + per)
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
    want = np.stack([np.concatenate([m.get_group(f, g)
                                     for f in range(len(osh))])
                     for g in range(len(m))])
    assert np.array_equal(want, groups[0])


def test_merged_float_statistics_over_ranks(tmp_path):
    """float_stats = 1: the ranks exchange binary64 sums instead of rows;
    the replicas stay bit-identical and the statistics are those of the rows
    (three ranks, ragged shards)"""
    import workloads
    world, N, config = 3, 5501, "gp_nich"
    mp.spawn(worker, args=(world, free_port(), str(tmp_path), config, 1, N, 1),
             nprocs=world, join=True)
    assign = np.concatenate([np.load(tmp_path / ("assign_%d.npy" % r))
                             for r in range(world)])
    counts = [np.load(tmp_path / ("counts_%d.npy" % r)) for r in range(world)]
    groups = [np.load(tmp_path / ("groups_%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(counts[0], counts[r])
        assert np.array_equal(groups[0], groups[r])
    _, _, vals, _ = workloads.make(config, N, K)
    # groups[0][g] = GP words (count, sum, log_prod) + NICH words (count,
    # mean, ctv); global ids are dense here up to removed groups: match the
    # groups by size and mean instead of by id
    x = vals[1].astype(np.float64)
    ids, sizes = np.unique(assign, return_counts=True)
    nich = groups[0][:, 3:6]
    got = sorted((int(np.int32(w[0])), float(w[1:2].view(np.float32)[0]))
                 for w in nich if np.int32(w[0]) > 0)
    want = sorted((int(c), float(x[assign == i].mean()))
                  for i, c in zip(ids, sizes))
    assert [g[0] for g in got] == [w[0] for w in want]
    np.testing.assert_allclose([g[1] for g in got], [w[1] for w in want],
                               rtol=1e-4, atol=1e-4)
