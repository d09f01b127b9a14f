"""The library's own RCCL communicator (dist_comm_*, dist_gibbs_sweep_sharded):
the sub-sweep loop runs inside the library with the all-reduce of the integer
delta image on the engine's stream.  One GPU on the test box, so the
communicator has one rank; what is checked is that the native loop is the same
function of the data as the host-driven loop: bit-identical assignments, group
sizes and statistics to the CPU oracle's batched sweep."""
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

K, BATCH, SWEEPS, SEED = 24, 1500, 3, 4242


def worker(rank, port, out, config, mode, N, peek=False, float_stats=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import oracle_lib as ol
    import workloads
    from distributions_amd import _core, engine
    dev = torch.device("cuda", 0)
    osh, gsh, vals, assign = workloads.make(config, N, K)
    cols = [torch.from_numpy(ol.value_words(s.kind, v).view(np.int32)
                             .copy()).to(dev) for s, v in zip(osh, vals)]
    packed = torch.from_numpy(assign.view(np.int32).copy()).to(dev)
    gpu = engine.Gibbs(1.0, 0.2, gsh)
    gpu.set_option("value_sorted", mode)
    gpu.set_option("device_normalise", 1)
    gpu.set_option("float_stats", float_stats)
    gpu.load_rows_torch(cols, packed.clone(), K, 2)
    sharded = engine.ShardedGibbs(gpu.core, N, 0, device=dev,
                                  force_collective=True,
                                  columns=cols, assign_packed=packed)
    sharded.sync_initial_stats()
    native = sharded.use_native_comm()
    for s in range(SWEEPS):
        sharded.sweep(BATCH, _core.rng_seed(SEED), draw_base=s * N)
        if peek:
            # a look at the state between two passes settles the open run;
            # the next pass takes it up again with the same bound on the
            # group count and the same batches left -- no word between the
            # ranks (a rank that never looked goes on with its run and issues
            # the very same collectives)
            assert len(gpu) >= K
            assert gpu.validate()["code"] == 0
    torch.cuda.synchronize()
    np.save(os.path.join(out, "resumed.npy"),
            np.array([gpu.core.debug_counts()["resumed_runs"]]))
    np.save(os.path.join(out, "native.npy"), np.array([int(native)]))
    np.save(os.path.join(out, "on_device.npy"),
            np.array([gpu.core.debug_counts()["device_normalised"]]))
    np.save(os.path.join(out, "assign.npy"), gpu.assignments())
    np.save(os.path.join(out, "counts.npy"), gpu.counts())
    np.save(os.path.join(out, "groups.npy"), np.stack([
        np.concatenate([gpu.get_group(f, g) for f in range(len(gsh))])
        for g in range(len(gpu))]))
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("config,mode,N,peek", [
    ("dd", 2, 9001, False), ("dd", 2, 9001, True), ("dd", 0, 9001, False),
    ("bb", 1, 9001, False), ("dd_bb_gp", 1, 6000, False)])
def test_native_loop_equals_oracle(tmp_path, config, mode, N, peek):
    import oracle_lib as ol
    import workloads
    mp.spawn(worker, args=(free_port(), str(tmp_path), config, mode, N, peek),
             nprocs=1, join=True)
    osh, gsh, vals, assign = workloads.make(config, N, K)
    native = bool(np.load(tmp_path / "native.npy")[0])
    # NormalInverseChiSq statistics and GammaPoisson's log-product depend on
    # the order of the adds: such engines exchange rows, not an integer image,
    # and the native loop must decline them (torch.distributed path is used)
    ordered = any(s.kind in (ol.NICH, ol.GP) for s in osh)
    assert native == (not ordered)
    # the value-sorted single-feature passes also normalise the group set on
    # the device (no host round trip between the all-reduces)
    on_device = int(np.load(tmp_path / "on_device.npy")[0])
    assert (on_device > 0) == (native and len(osh) == 1 and mode == 2)
    # a run closed by a look at the state is taken up again, pass after pass
    resumed = int(np.load(tmp_path / "resumed.npy")[0])
    assert resumed == ((SWEEPS - 1) if peek and on_device else 0)
    m = ol.OracleMixture(1.0, 0.2, osh)
    m.init_from_assignments(vals, assign, K, 2)
    for s in range(SWEEPS):
        for b in range(0, N, BATCH):
            m.gibbs_batch(b, min(N, b + BATCH), ol.oracle().orc_rng_seed(SEED),
                          s * N)
    assert np.array_equal(np.load(tmp_path / "assign.npy"), m.assign)
    assert np.array_equal(np.load(tmp_path / "counts.npy"), m.counts())
    want = np.stack([np.concatenate([m.get_group(f, g)
                                     for f in range(len(osh))])
                     for g in range(len(m))])
    assert np.array_equal(np.load(tmp_path / "groups.npy"), want)


def test_native_loop_takes_merged_float_statistics(tmp_path):
    """BASELINE configs[2]'s feature list (GammaPoisson + NormalInverseChiSq)
    with float_stats = 1: the order-dependent statistics travel as binary64
    sums (one more all-reduce per sub-sweep), so the library's own loop takes
    the engine instead of handing it back to the Python loop -- and is the
    same function of the data as the single engine with the same option."""
    import workloads
    from distributions_amd import _core, engine
    config, N = "gp_nich", 6000
    mp.spawn(worker, args=(free_port(), str(tmp_path), config, 1, N, False, 1),
             nprocs=1, join=True)
    assert bool(np.load(tmp_path / "native.npy")[0])
    osh, gsh, vals, assign = workloads.make(config, N, K)
    gpu = engine.Gibbs(1.0, 0.2, gsh)
    gpu.set_option("value_sorted", 1)
    gpu.set_option("float_stats", 1)
    gpu.load_rows(vals, assign, K, 2)
    for s in range(SWEEPS):
        gpu.sweep(0, N, BATCH, SEED, draw_base=s * N)
    assert gpu.core.debug_counts()["merged_batches"] > 0
    assert np.array_equal(np.load(tmp_path / "assign.npy"), gpu.assignments())
    assert np.array_equal(np.load(tmp_path / "counts.npy"), gpu.counts())
    want = np.stack([np.concatenate([gpu.get_group(f, g)
                                     for f in range(len(gsh))])
                     for g in range(len(gpu))])
    assert np.array_equal(np.load(tmp_path / "groups.npy"), want)
