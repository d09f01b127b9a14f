"""rows/s of the exact sequential chain (batches of one) on the headline
workload: what a caller pays for reference-identical sequential semantics"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributions_amd import _core, engine
dev = torch.device("cuda", 0)
n, k, dim = 1_000_000, 1024, 256
gen = torch.Generator(device=dev)
gen.manual_seed(20240601)
values = torch.randint(0, dim, (n,), generator=gen, device=dev,
                       dtype=torch.int32)
assign = torch.arange(n, device=dev, dtype=torch.int64).remainder(k).to(
    torch.int32)
g = engine.Gibbs(1.0, 0.2, [engine.dd_shared([0.5] * dim)])
g.load_rows_torch([values], assign, k, 1)
st = _core.rng_seed(1)
st = g.sweep_sequential(0, 200, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
rows = 3000
st = g.sweep_sequential(200, 200 + rows, st)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("sequential chain: %.0f rows/s (%.1f us/row)" % (rows / dt,
                                                       dt / rows * 1e6))
