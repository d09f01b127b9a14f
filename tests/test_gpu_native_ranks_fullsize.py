"""BASELINE configs[3] and configs[4] in their real shapes on the ranks of
ONE GPU (host transport): 8 value-partitioned ranks of 1.25M rows each --
DirichletDiscrete(256), K = 1024, sub-sweeps of 125 000 rows per rank (10^6
in all), and DirichletProcessDiscrete(V = 10 000), K = 8192 -- through
dist_gibbs_sweep_sharded.  The oracle cannot follow 10^7 rows through the
naive batch loop in test time; what is held here is what any replica-exchange
bug breaks at once: every rank ends with the same group sizes and -- after
gather_cells -- the same statistics word for word, the sizes are those of the
rows' assignments summed over ranks, and no exchange carried more than
4 + 3 * (bound on the live group count) words."""
import hashlib
import os
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

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

WORLD, PER_RANK, SEED = 8, 1_250_000, 99


def worker(rank, world, port, out, config, dim, k, per, sweeps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("DIST_COMM_TIMEOUT_S", "300")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distributions_amd import _core, engine
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n = PER_RANK
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED + rank)
    lo, hi = rank * dim // world, (rank + 1) * dim // world
    col = torch.randint(lo, hi, (n,), generator=gen, device=dev,
                        dtype=torch.int32)
    packed = (torch.arange(n, device=dev, dtype=torch.int64)
              + rank * n).remainder(k).to(torch.int32)
    shared = (engine.dd_shared([0.5] * dim) if config == "dd"
              else engine.dpd_shared(0.5, [1.0 / dim] * dim, 0.0))
    gpu = engine.Gibbs(1.0, 0.2, [shared])
    gpu.set_option("device_normalise", 1)
    gpu.load_rows_torch([col], packed.clone(), k, 1, row_offset=rank * n)
    sharded = engine.ShardedGibbs(gpu.core, n, rank * n, device=dev,
                                  columns=[col], assign_packed=packed)
    sharded.sync_initial_stats()
    assert sharded.use_native_comm()
    sharded.partition_by_value()
    for s in range(sweeps):
        sharded.sweep(per, _core.rng_seed(SEED), draw_base=s * n * world)
    torch.cuda.synchronize()
    vol = gpu.core.comm_volume()
    dbg = gpu.core.debug_counts()
    sharded.gather_cells()
    counts = gpu.counts()
    assign = gpu.assignments()
    h = hashlib.sha256()
    for g in range(0, len(gpu), max(1, len(gpu) // 64)):   # 64 groups' words
        h.update(gpu.get_group(0, g).tobytes())
    np.save(os.path.join(out, "counts_%d.npy" % rank), counts)
    np.save(os.path.join(out, "local_hist_%d.npy" % rank),
            np.bincount(assign, minlength=int(assign.max()) + 1))
    np.save(os.path.join(out, "gids_%d.npy" % rank), np.array(
        [gpu.core.packed_to_global(g) for g in range(len(gpu))], np.int64))
    with open(os.path.join(out, "hash_%d.txt" % rank), "w") as f:
        f.write(h.hexdigest())
    np.save(os.path.join(out, "meta_%d.npy" % rank), np.array(
        [vol["collectives"], vol["words_max"], dbg["device_normalised"],
         dbg["stream_batches"]], np.int64))
    dist.destroy_process_group()


@pytest.mark.parametrize("config,dim,k,sweeps", [
    ("dd", 256, 1024, 2),          # BASELINE configs[3] on configs[1]'s model
    ("dpd", 10000, 8192, 1)])      # BASELINE configs[4]: C5's 8-GPU leg
def test_eight_value_partitioned_ranks_at_baseline_shape(tmp_path, config,
                                                          dim, k, sweeps):
    from test_gpu_native_ranks import free_port
    per = 125_000
    mp.spawn(worker, args=(WORLD, free_port(), str(tmp_path), config, dim, k,
                           per, sweeps), nprocs=WORLD, join=True)
    counts = [np.load(tmp_path / ("counts_%d.npy" % r)) for r in range(WORLD)]
    hashes = [open(tmp_path / ("hash_%d.txt" % r)).read()
              for r in range(WORLD)]
    gids = [np.load(tmp_path / ("gids_%d.npy" % r)) for r in range(WORLD)]
    for r in range(1, WORLD):
        assert np.array_equal(counts[0], counts[r])   # replicas agree ...
        assert np.array_equal(gids[0], gids[r])       # ... on the id maps ...
        assert hashes[0] == hashes[r]                 # ... and the statistics
    assert counts[0].sum() == WORLD * PER_RANK
    # the group sizes are those of the rows' assignments, summed over ranks
    size = max(len(np.load(tmp_path / ("local_hist_%d.npy" % r)))
               for r in range(WORLD))
    hist = np.zeros(size, np.int64)
    for r in range(WORLD):
        h = np.load(tmp_path / ("local_hist_%d.npy" % r))
        hist[:len(h)] += h
    by_gid = np.zeros(size, np.int64)
    live = gids[0][gids[0] < size]
    by_gid[live] = counts[0][gids[0] < size]
    assert np.array_equal(hist, by_gid)
    batches = sweeps * (PER_RANK // per)
    for r in range(WORLD):
        meta = np.load(tmp_path / ("meta_%d.npy" % r))
        assert meta[0] == batches and meta[2] == batches
        # SURVEY 8(e) / VERDICT: in-run words <= 3 * K_bound (+ the header)
        assert meta[1] <= 4 + 3 * (k + 1 + batches)
