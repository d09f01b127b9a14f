import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributions_amd import _core, engine
dev = torch.device("cuda", 0)
n, k, dim = 10_000_000, 1024, 256
gen = torch.Generator(device=dev); gen.manual_seed(20240601)
values = torch.randint(0, dim, (n,), generator=gen, device=dev, dtype=torch.int32)
assign = torch.arange(n, device=dev, dtype=torch.int64).remainder(k).to(torch.int32)
g = engine.Gibbs(1.0, 0.2, [engine.dd_shared([0.5] * dim)])
g.load_rows_torch([values], assign, k, 1)
st = _core.rng_seed(1)
for i in range(26):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.core.sweep(0, n, 1_000_000, st, i * n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, launches, rows = g.kernel_stats(reset=True)
    print("sweep %2d  %.3f ms  kernel %.3f ms  K=%d" % (i, dt * 1e3, ms, len(g)))
