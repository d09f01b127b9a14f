"""Two real GPUs, two ranks, RCCL: the sharded sweep (library-owned
communicator, dist_gibbs_sweep_sharded) equals the single-rank run with the
same batch composition bit for bit.  Skipped on a one-GPU box (there the same
engine code runs as two and three ranks sharing the GPU with collectives
staged through gloo, tests/test_gpu_two_ranks.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_count():
    import torch
    return torch.cuda.device_count()


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import workloads
from distributions_amd import _core, engine
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("nccl", rank=rank, world_size=world)
torch.cuda.set_device(rank); _core.set_device(rank)
dev = torch.device("cuda", rank)
n, k, batch, sweeps = 240000, 96, 30000, 2
osh, gsh, vals, assign = workloads.make("dd", n, k, dim=64)
lo, hi = rank * n // world, (rank + 1) * n // world
g = engine.Gibbs(1.0, 0.2, gsh)
g.load_rows([vals[0][lo:hi]], assign[lo:hi], k, 1, row_offset=lo)
sh = engine.ShardedGibbs(g.core, hi - lo, lo, device=dev)
sh.sync_initial_stats()
assert sh.use_native_comm()
st = _core.rng_seed(11)
for s in range(sweeps):
    sh.sweep(batch, st, draw_base=s * n)
np.save(os.path.join(%(out)r, "assign_%%d.npy" %% rank), g.assignments())
np.save(os.path.join(%(out)r, "counts_%%d.npy" %% rank), g.counts())
dist.destroy_process_group()
'''


@pytest.mark.skipif(gpu_count() < 2, reason="needs two GPUs")
def test_two_rank_rccl_sweep_equals_the_oracle(tmp_path):
    import oracle_lib as ol
    import workloads
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path)})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    assert [p.wait(timeout=900) for p in procs] == [0, 0]
    # the oracle with the same batch composition: per sub-sweep the rows
    # [b, b + 30000) of BOTH shards against one snapshot
    n, k, batch, sweeps, world = 240000, 96, 30000, 2, 2
    osh, gsh, vals, assign = workloads.make("dd", n, k, dim=64)
    orc = ol.OracleMixture(1.0, 0.2, osh)
    orc.init_from_assignments(vals, assign, k, 1)
    back = ol.OracleBackend(orc, 0)
    st = ol.oracle().orc_rng_seed(11)
    half = n // world
    L = orc.L
    ol._phase_sigs(L)
    for s in range(sweeps):
        for b in range(0, half, batch):
            snap = orc.counts().copy()
            moves = []
            for r in range(world):
                r0, r1 = r * half + b, r * half + min(half, b + batch)
                old = np.zeros(r1 - r0 + 1, np.uint32)
                new = np.zeros(r1 - r0 + 1, np.uint32)
                L.orc_mix_batch_sample(orc.h, r0, r1, orc._vals, orc.assign,
                                       st, s * n, 0, old, new)
                moves.append((r0, r1, old, new))
            for r0, r1, old, new in moves:
                L.orc_mix_apply_moves(orc.h, r0, r1, orc._vals, orc.assign,
                                      old, new)
            L.orc_mix_batch_finish(orc.h, np.ascontiguousarray(snap, np.int32))
    got = np.concatenate([np.load(tmp_path / ("assign_%d.npy" % r))
                          for r in range(world)])
    np.testing.assert_array_equal(got, orc.assign)
    for r in range(world):
        np.testing.assert_array_equal(
            np.load(tmp_path / ("counts_%d.npy" % r)), orc.counts())
