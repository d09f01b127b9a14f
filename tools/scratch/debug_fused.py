import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import oracle_lib as ol
import workloads
from distributions_amd import engine

def run(per_batch, fused, batch=256, n=4096, k=32, config="dd", sweeps=3):
    osh, gsh, vals, assign = workloads.make(config, n, k)
    orc = ol.OracleMixture(1.0, 0.2, osh)
    orc.init_from_assignments(vals, assign, k, 1)
    gpu = engine.Gibbs(1.0, 0.2, gsh)
    gpu.set_option("value_sorted", 2)
    gpu.set_option("value_stream", 0)
    gpu.set_option("narrow_tiles", 0)
    gpu.set_option("fused_tables", fused)
    gpu.load_rows(vals, assign, k, 1)
    seed = 12345
    st = ol.oracle().orc_rng_seed(seed)
    for sweep in range(sweeps):
        base = sweep * n
        for b in range(0, n, batch):
            pre_assign = orc.assign.copy(); pre_counts = orc.counts().copy()
            pre_g2p = [int(ol.oracle().orc_mix_global_to_packed(orc.h, int(a))) for a in pre_assign[b:b+batch]]
            orc.gibbs_batch(b, min(n, b + batch), st, base)
            if per_batch:
                gpu.sweep(b, min(n, b + batch), batch, seed, draw_base=base)
                got, want = gpu.assignments(), orc.assign
                bad = np.nonzero(got != want)[0]
                if len(gpu) != len(orc) or bad.size:
                    print("per_batch", per_batch, "fused", fused, "sweep", sweep, "batch", b,
                          "len", len(gpu), len(orc), "bad rows", bad[:10],
                          got[bad[:10]], want[bad[:10]])
                    oc = orc.counts(); gc = gpu.counts()
                    print(" counts gpu", gc[:60]); print(" counts orc", oc[:60])
                    for r in bad[:10]:
                        pg = pre_g2p[r - b]
                        print("  row", r, "value", vals[0][r], "old packed", pg, "old size", pre_counts[pg], "K before", len(pre_counts))
                    return
        if not per_batch:
            gpu.sweep(0, n, batch, seed, draw_base=base)
            got, want = gpu.assignments(), orc.assign
            bad = np.nonzero(got != want)[0]
            print("sweep", sweep, "len", len(gpu), len(orc), "bad", bad.size, bad[:10],
                  got[bad[:10]], want[bad[:10]])
            if bad.size or len(gpu) != len(orc):
                oc = orc.counts(); gc = gpu.counts()
                print(" counts gpu", gc); print(" counts orc", oc)
                # singleton rows at the start of this sweep?
                return
    print("ok per_batch", per_batch, "fused", fused, gpu.core.debug_counts())

run(1, 1)
run(1, 1, sweeps=3, k=32, batch=512)
