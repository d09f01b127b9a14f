"""Independent exact chains on one GPU: T host threads, each with its own HIP
stream and engine, run the reference's sequential chain concurrently
(dist_set_stream).  Prints the aggregate rate and checks that chains with the
same seed and rows agree.  usage: chains.py [threads] [rows]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distributions_amd import _core, engine

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
n, k, dim = 100_000, 1024, 256
rng = np.random.default_rng(1)
values = rng.integers(0, dim, n).astype(np.uint32)
assign = (np.arange(n) % k).astype(np.uint32)
results = [None] * T
barrier = threading.Barrier(T + 1)


def worker(t):
    stream = torch.cuda.Stream()
    _core.set_stream(stream.cuda_stream)
    g = engine.Gibbs(1.0, 0.2, [engine.dd_shared([0.5] * dim)])
    g.load_rows([values], assign, k, 1)
    st = g.sweep_sequential(0, 100, _core.rng_seed(7))     # warm
    barrier.wait()
    st = g.sweep_sequential(100, 100 + ROWS, st)
    _core.synchronize()
    barrier.wait()
    results[t] = (st, g.assignments()[:100 + ROWS].copy())
    _core.set_stream(0)


threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
for th in threads:
    th.start()
barrier.wait()
t0 = time.perf_counter()
barrier.wait()
dt = time.perf_counter() - t0
for th in threads:
    th.join()
same = all(r[0] == results[0][0] and np.array_equal(r[1], results[0][1])
           for r in results)
print("%d chains x %d rows: %.0f rows/s aggregate (%.1f us per row and chain); "
      "identical chains agree: %s" % (T, ROWS, T * ROWS / dt,
                                       dt / ROWS * 1e6, same))
