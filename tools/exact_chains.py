"""Rate of the exact sequential chain (DirichletDiscrete(256), K = 1024 + 1,
BASELINE configs[1]'s model): one chain through k_chains (structural steps on
the device) and through round 3's kernel, and M independent chains in one
launch (dist_gibbs_sweep_sequential_many; BASELINE configs[3]).
usage: exact_chains.py [rows per chain] [M ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distributions_amd import _core, engine

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
MS = [int(a) for a in sys.argv[2:]] or [8, 64, 256, 512, 1024]
n, k, dim = max(ROWS + 200, 20000), 1024, 256


def chain(seed, mode=2):
    rng = np.random.default_rng(seed)
    values = rng.integers(0, dim, n).astype(np.uint32)
    assign = (np.arange(n) % k).astype(np.uint32)
    g = engine.Gibbs(1.0, 0.2, [engine.dd_shared([0.5] * dim)])
    g.set_option("debug.sequential_chain", mode)
    g.load_rows([values], assign, k, 1)
    return g


for mode, name in ((2, "k_chains"), (1, "round 3's kernel (k_chain_rows)")):
    g = chain(1, mode)
    st = g.sweep_sequential(0, 100, _core.rng_seed(7))   # warm
    _core.synchronize()
    t0 = time.perf_counter()
    st = g.sweep_sequential(100, 100 + ROWS, st)
    _core.synchronize()
    dt = time.perf_counter() - t0
    print("1 chain, %-32s %9.0f rows/s  (%.2f us per row = %.0f cycles at "
          "2.4 GHz; %d launches)" % (name + ":", ROWS / dt, dt / ROWS * 1e6,
                                     dt / ROWS * 2.4e9,
                                     g.core.chain_launches()), flush=True)
    del g

for m in MS:
    gs = [chain(100 + i) for i in range(m)]
    states = np.array([_core.rng_seed(9000 + i) for i in range(m)], np.uint32)
    states = _core.sweep_sequential_many([g.core for g in gs], 0, 100, states)
    _core.synchronize()
    t0 = time.perf_counter()
    states = _core.sweep_sequential_many([g.core for g in gs], 100,
                                         100 + ROWS, states)
    _core.synchronize()
    dt = time.perf_counter() - t0
    print("%4d chains in one launch: %12.0f rows/s aggregate  (%.2f us per "
          "row and chain)" % (m, m * ROWS / dt, dt / ROWS * 1e6), flush=True)
    del gs
