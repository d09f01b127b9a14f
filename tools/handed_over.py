"""How many rows of the headline workload are alone in their group (the rows
k_vs_sample hands over to k_vs_apply), sweep by sweep:
python tools/handed_over.py [sweeps] [rows] [batch]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from distributions_amd import _core, engine

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
k, dim = 1024, 256
dev = torch.device("cuda", 0)
_core.set_device(0)
gen = torch.Generator(device=dev)
gen.manual_seed(20240601)
values = torch.randint(0, dim, (n,), generator=gen, device=dev,
                       dtype=torch.int32)
assign = torch.arange(n, device=dev, dtype=torch.int64).remainder(k).to(
    torch.int32)
shared = engine.dd_shared([0.5] * dim)
g = engine.Gibbs(1.0, 0.2, [shared])
g.set_option("value_sorted", 1)
g.set_option("device_normalise", 2)
g.load_rows_torch([values], assign, k, 1)
seed = _core.rng_seed(20240601)
for s in range(sweeps):
    g.sweep(0, n, batch, seed, draw_base=s * n)
    c = np.asarray(g.counts())
    print("sweep %2d: %d groups, %d of one row, %d of two, smallest ten %s"
          % (s, len(c), int((c == 1).sum()), int((c == 2).sum()),
             np.sort(c)[:10].tolist()), flush=True)
