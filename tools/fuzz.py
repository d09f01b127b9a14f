"""Differential fuzzing of the engine against the oracle: random feature lists,
hyper-parameters, clustering models, group sets, batch tilings, kernel choices
and interleaved sequential stretches.  usage: fuzz.py [trials] [first_seed]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from distributions_amd import engine  # noqa: E402


def same_state(orc, gpu):
    # the engine's own consistency first (dist_gibbs_validate: a recount from
    # the rows on the device): independent of the oracle
    report = gpu.validate(raise_on_failure=False)
    if report["code"]:
        return "validate: %r" % (report,)
    if len(gpu) != len(orc):
        return "group count %d vs %d" % (len(gpu), len(orc))
    if not np.array_equal(gpu.counts(), orc.counts()):
        return "counts"
    if not np.array_equal(gpu.assignments(), orc.assign):
        bad = np.nonzero(gpu.assignments() != orc.assign)[0]
        return "assignment of row %d" % bad[0]
    for f in range(orc.F):
        for g in range(len(orc)):
            if not np.array_equal(gpu.get_group(f, g), orc.get_group(f, g)):
                return "statistics of feature %d group %d" % (f, g)
    return None


def trial(seed, large=False):
    rng = np.random.default_rng(seed)
    L = ol.oracle()
    L.orc_mix_set_low_entropy.restype = None
    L.orc_mix_set_low_entropy.argtypes = [ctypes.c_void_p, ctypes.c_int]
    n = int(rng.choice([1, 2, 63, 64, 65, 500, 2000, 5000]))
    k = int(min(n, rng.choice([1, 2, 7, 33, 150])))
    if large:   # group sets that cross the kernels' capacity steps
        n = int(rng.choice([20000, 60000]))
        k = int(rng.choice([250, 1000, 3000, 8192]))
        if k == 8192:   # (the oracle does 12 k row-updates/s there)
            n = 20000
    empty = int(rng.integers(1, 4))
    nf = int(rng.choice([1, 1, 1, 2, 3]))
    feats_o, feats_g, vals, desc = [], [], [], []
    for _ in range(nf):
        kind = rng.choice(["dd", "bb", "gp", "nich", "bnb", "dpd"])
        if large and rng.random() < 0.4:
            kind = "dpd"
        desc.append(str(kind))
        if kind == "dd":
            dim = int(rng.choice([1, 2, 16, 256]))
            alphas = [float(a) for a in rng.uniform(0.05, 3, dim)]
            feats_o.append(ol.make_shared(ol.DD, alphas=alphas))
            feats_g.append(engine.dd_shared(alphas))
            vals.append(rng.integers(0, dim, n).astype(np.uint32))
        elif kind == "dpd":
            dim = int(rng.choice([3, 40, 700]))
            if large:   # BASELINE configs[4]'s table: up to 10 000 values
                dim = int(rng.choice([40, 700, 3000, 10000]))
            betas = rng.dirichlet(np.ones(dim)).astype(np.float32) * 0.9
            feats_o.append(ol.make_shared(ol.DPD, alpha=0.7, betas=betas,
                                          beta0=0.1))
            feats_g.append(engine.dpd_shared(0.7, betas, 0.1))
            vals.append(rng.integers(0, dim, n).astype(np.uint32))
        elif kind == "bb":
            a, b = float(rng.uniform(0.1, 3)), float(rng.uniform(0.1, 3))
            feats_o.append(ol.make_shared(ol.BB, alpha=a, beta=b))
            feats_g.append(engine.bb_shared(a, b))
            vals.append((rng.random(n) < rng.uniform(0, 1)).astype(np.uint32))
        elif kind == "gp":
            a, ib = float(rng.uniform(0.2, 4)), float(rng.uniform(0.2, 4))
            feats_o.append(ol.make_shared(ol.GP, alpha=a, inv_beta=ib))
            feats_g.append(engine.gp_shared(a, ib))
            v = rng.poisson(float(rng.uniform(0.3, 40)), n).astype(np.uint32)
            if rng.random() < 0.3:
                v[rng.integers(0, n, max(1, n // 50))] = rng.integers(
                    256, 10 ** 6)
            vals.append(v)
        elif kind == "bnb":
            a, b = float(rng.uniform(0.2, 3)), float(rng.uniform(0.2, 3))
            r = int(rng.integers(1, 5))
            feats_o.append(ol.make_shared(ol.BNB, alpha=a, beta=b, r=r))
            feats_g.append(engine.bnb_shared(a, b, r))
            vals.append(rng.negative_binomial(r, 0.3, n).astype(np.uint32))
        else:
            p = [float(rng.normal()), float(rng.uniform(0.1, 3)),
                 float(rng.uniform(0.1, 3)), float(rng.uniform(0.02, 5))]
            feats_o.append(ol.make_shared(ol.NICH, mu=p[0], kappa=p[1],
                                          sigmasq=p[2], nu=p[3]))
            feats_g.append(engine.nich_shared(*p))
            vals.append((rng.normal(size=n) * rng.uniform(0.01, 100)).astype(
                np.float32))
    assign = rng.integers(0, k, n).astype(np.uint32)
    assign[:k] = np.arange(k)
    low_entropy = rng.random() < 0.25
    alpha, d = float(rng.uniform(0.1, 10)), float(rng.uniform(0, 0.9))
    orc = ol.OracleMixture(alpha, d, feats_o)
    if low_entropy:
        ds = n + int(rng.choice([0, 1, 1000]))
        L.orc_mix_set_low_entropy(orc.h, ds)
        gpu = engine.Gibbs(alpha, d, feats_g, dataset_size=ds)
    else:
        gpu = engine.Gibbs(alpha, d, feats_g)
    orc.init_from_assignments(vals, assign, k, empty)
    mode = int(rng.choice([0, 1, 2]))
    gpu.set_option("value_sorted", mode)
    if rng.integers(0, 2):   # the per-value running sums, also on small launches
        gpu.set_option("debug.running_sums_min_tiles", 0)
    # per-value tables or the table-free kernel (and its packed apply chunks)
    gpu.set_option("value_stream", int(rng.choice([0, 1, 2])))
    gpu.set_option("narrow_tiles", int(rng.choice([0, 1, 2])))
    gpu.set_option("debug.narrow_read_ahead", int(rng.choice([0, 4, 8])))
    gpu.set_option("debug.stream_scratch", int(rng.choice([0, 1])))
    gpu.set_option("device_normalise", int(rng.choice([0, 1, 2])))
    gpu.set_option("fused_tables", int(rng.choice([0, 1, 1, 1])))
    # the general-row kernels: round 2's, k_rows_scratch with the likelihoods
    # / scores as well in its scratch, without a scratch; LDS or global
    # FastLog table; workgroup size; folded leading features; staged apply
    gpu.set_option("debug.rows_scratch", int(rng.choice([0, 3, 3, 3])))
    gpu.set_option("debug.rows_scratch_lds_log", int(rng.choice([0, 1])))
    gpu.set_option("debug.rows_scratch_block", int(rng.choice([64, 256, 512, 1024])))
    gpu.set_option("debug.rows_fold", int(rng.choice([0, 1, 2, 2])))
    gpu.set_option("debug.apply_stage", int(rng.choice([0, 1])))
    gpu.set_option("debug.program_all", int(rng.choice([0, 1, 1])))
    gpu.load_rows(vals, assign, k, empty)
    what = "seed %d: n=%d k=%d empty=%d feats=%s mode=%d %s" % (
        seed, n, k, empty, "+".join(desc), mode,
        "LowEntropy" if low_entropy else "PY(%.2f,%.2f)" % (alpha, d))
    err = same_state(orc, gpu)
    if err:
        return what + " after load: " + err
    eng_seed = int(rng.integers(1, 2 ** 31 - 1))
    st = L.orc_rng_seed(eng_seed)
    draw = 0
    for step in range(4):
        if rng.random() < 0.3 and n > 1:
            a = int(rng.integers(0, n))
            b = int(min(n, a + rng.integers(1, 200)))
            state = orc.gibbs_sequential(a, b, st)
            got = gpu.sweep_sequential(a, b, st)
            if got != state:
                return what + " step %d sequential: engine state" % step
            # the chain consumed its own stream; batches keep theirs
        else:
            batch = int(rng.choice([1, 7, 64, 333, 4096, n]))
            if large:
                batch = int(rng.choice([4096, 20000, n]))
            for b0 in range(0, n, batch):
                orc.gibbs_batch(b0, min(n, b0 + batch), st, draw)
            gpu.sweep(0, n, batch, eng_seed, draw_base=draw)
            draw += n
        # (not after every step: a device-normalised run stays open from one
        # sweep to the next until somebody looks)
        if step == 3 or rng.random() < 0.5:
            err = same_state(orc, gpu)
            if err:
                return what + " step %d: %s" % (step, err)
    return None


def trial_collective(seed):
    """the multi-GPU driver (integer deltas + all-reduce, ordered statistics
    gathered and replayed) on ONE rank against the oracle's plain batches;
    needs an initialised process group"""
    import torch
    rng = np.random.default_rng(seed)
    L = ol.oracle()
    n = int(rng.choice([64, 500, 3000]))
    k = int(rng.choice([2, 9, 40]))
    config = str(rng.choice(["dd", "bb", "gp", "nich", "gp_nich", "bnb",
                             "dd_bb_gp"]))
    import workloads
    osh, gsh, vals, assign = workloads.make(config, n, k, seed=seed)
    alpha, d = float(rng.uniform(0.2, 5)), float(rng.uniform(0, 0.8))
    orc = ol.OracleMixture(alpha, d, osh)
    orc.init_from_assignments(vals, assign, k, 2)
    dev = torch.device("cuda", 0)
    cols = [torch.from_numpy(ol.value_words(s.kind, v).view(np.int32)).to(dev)
            for s, v in zip(osh, vals)]
    a = torch.from_numpy(assign.view(np.int32)).to(dev)
    gpu = engine.Gibbs(alpha, d, gsh)
    gpu.set_option("value_sorted", int(rng.choice([0, 1, 2])))
    if rng.integers(0, 2):
        gpu.set_option("debug.running_sums_min_tiles", 0)
    gpu.set_option("value_stream", int(rng.choice([0, 1, 2])))
    gpu.set_option("narrow_tiles", int(rng.choice([0, 1, 2])))
    gpu.set_option("debug.narrow_read_ahead", int(rng.choice([0, 4, 8])))
    gpu.set_option("debug.stream_scratch", int(rng.choice([0, 1])))
    gpu.set_option("fused_tables", int(rng.choice([0, 1, 1])))
    gpu.set_option("debug.rows_scratch", int(rng.choice([0, 3])))
    gpu.set_option("debug.rows_fold", int(rng.choice([0, 1, 2])))
    gpu.set_option("debug.apply_stage", int(rng.choice([0, 1])))
    gpu.load_rows_torch(cols, a.clone(), k, 2)
    sharded = engine.ShardedGibbs(gpu.core, n, 0, device=dev,
                                  force_collective=True, columns=cols,
                                  assign_packed=a)
    sharded.sync_initial_stats()
    # half of the trials run the sub-sweep loop inside the library, on its
    # own RCCL communicator (one per process, shared by the engines)
    global _NATIVE_COMM
    native = False
    if rng.integers(0, 2):
        native = sharded.use_native_comm(_NATIVE_COMM)
        if native:
            _NATIVE_COMM = sharded.native_comm
    what = "collective seed %d: %s n=%d k=%d native=%d" % (seed, config, n, k,
                                                          native)
    err = same_state(orc, gpu)
    if err:
        return what + " after the initial exchange: " + err
    eng_seed = int(rng.integers(1, 2 ** 31 - 1))
    st = L.orc_rng_seed(eng_seed)
    for sweep in range(3):
        batch = int(rng.choice([50, 400, n]))
        for b0 in range(0, n, batch):
            orc.gibbs_batch(b0, min(n, b0 + batch), st, sweep * n)
        sharded.sweep(batch, eng_seed, draw_base=sweep * n)
        torch.cuda.synchronize()
        err = same_state(orc, gpu)
        if err:
            return what + " sweep %d batch %d: %s" % (sweep, batch, err)
    return None


_NATIVE_COMM = None


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    collective = len(sys.argv) > 3 and sys.argv[3] == "collective"
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    if collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group("nccl", rank=0, world_size=1)
    failures = 0
    for seed in range(first, first + trials):
        try:
            err = (trial_collective(seed) if collective
                   else trial(seed, large=large))
        except Exception as e:   # noqa: BLE001
            err = "seed %d: exception %r" % (seed, e)
        if err:
            failures += 1
            print("FAIL", err, flush=True)
    print("%d trials, %d failures" % (trials, failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
