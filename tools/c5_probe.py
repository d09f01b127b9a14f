"""C5-shaped probe: DPD V=10000, K=8192, N rows, a few sub-sweeps; prints the
engine's path diagnostics (for use under rocprofv3)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from distributions_amd import _core, engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
k, dim = 8192, 10000
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
col = torch.randint(0, dim, (n,), generator=gen, device=dev, dtype=torch.int32)
assign = torch.arange(n, device=dev, dtype=torch.int64).remainder(k).to(torch.int32)
g = engine.Gibbs(1.0, 0.2, [engine.dpd_shared(0.5, [1.0 / dim] * dim, 0.0)])
g.set_option("value_stream", int(os.environ.get("STREAM", "1")))
g.load_rows_torch([col], assign, k, 1)
for s in range(sweeps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.sweep(0, n, batch, 5, draw_base=s * n)
    torch.cuda.synchronize()
    print("sweep %d: %.2f ms  %s" % (s, 1e3 * (time.perf_counter() - t0), g.core.debug_counts()), flush=True)
