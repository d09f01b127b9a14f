"""CPU-side cost of the calls ShardedGibbs.sweep makes per sub-sweep (one rank,
forced collective path): where the multi-GPU path spends host time"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from distributions_amd import _core, engine
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29588")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, k, dim, B = 10_000_000, 1024, 256, 1_000_000
gen = torch.Generator(device=dev)
gen.manual_seed(1)
values = torch.randint(0, dim, (n,), generator=gen, device=dev,
                       dtype=torch.int32)
assign = torch.arange(n, device=dev, dtype=torch.int64).remainder(k).to(
    torch.int32)
g = engine.Gibbs(1.0, 0.2, [engine.dd_shared([0.5] * dim)])
g.load_rows_torch([values], assign, k, 1)
core = g.core
st = _core.rng_seed(1)
names = ["sample", "alloc", "delta", "all_reduce", "apply", "finish"]
acc = dict.fromkeys(names, 0.0)
batches = 0
for sweep in range(6):
    torch.cuda.synchronize()
    t_sweep = time.perf_counter()
    for b in range(0, n, B):
        t = [time.perf_counter()]
        core.batch_sample(b, b + B, st, sweep * n)
        t.append(time.perf_counter())
        delta = torch.empty(core.stat_words(), dtype=torch.int32, device=dev)
        t.append(time.perf_counter())
        core.batch_delta_dev(int(delta.data_ptr()))
        t.append(time.perf_counter())
        dist.all_reduce(delta)
        t.append(time.perf_counter())
        core.batch_apply_delta_dev(int(delta.data_ptr()))
        t.append(time.perf_counter())
        core.batch_finish()
        t.append(time.perf_counter())
        if sweep >= 2:
            for i, name in enumerate(names):
                acc[name] += t[i + 1] - t[i]
            batches += 1
    torch.cuda.synchronize()
    print("sweep %d: %.3f ms" % (sweep, (time.perf_counter() - t_sweep) * 1e3))
for name in names:
    print("%-10s %.1f us per sub-sweep (host time)" % (
        name, acc[name] / batches * 1e6))
sharded = engine.ShardedGibbs(core, n, 0, device=dev, force_collective=True)
for sweep in range(6, 10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sharded.sweep(B, st, draw_base=sweep * n)
    torch.cuda.synchronize()
    print("ShardedGibbs sweep %d: %.3f ms" % (
        sweep, (time.perf_counter() - t0) * 1e3))
dist.destroy_process_group()
