#!/usr/bin/env python3
"""bench.py -- row-Gibbs-updates/s of the batched mixture row update.

Workload (BASELINE.json configs[1], SURVEY 8d "C2"): DirichletDiscrete
dim=256, alphas=0.5, K=1024 non-empty groups + 1 empty, N=10M rows per GPU,
values iid uniform{0..255}, initial assignment i mod K, PitmanYor(alpha=1,
d=0.2).  A step = one full Gibbs pass over the resident rows (self-remove,
score all K groups, sample, add), in frozen sub-sweeps of --batch rows, with
the statistics update, group-set normalisation and cache rebuild included.
Rows are generated on the device (seeded) before the timed region.

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL); rows
are sharded, weak scaling (N rows per GPU), one all-reduce of the integer
statistic deltas per sub-sweep.

Prints ONE JSON line (rank 0).  `roofline` prices the score+sample kernel
(k_vs_sample, or k_sweep_sample when the value-sorted kernel does not apply)
by ALGORITHMIC bytes: (12*K + 12) B per row over its HIP-event
duration on the launch stream; `traffic` is the HBM byte count per launch from
the committed rocprofv3 PMC passes (profiles/).  `cpu_baseline` times the
oracle's sequential chain (the reference loop restated, oracle/oracle.c) on a
bounded row sample of the same workload, one host thread.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000,
                    help="rows per GPU")
    ap.add_argument("--groups", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1_000_000,
                    help="rows per frozen sub-sweep (per GPU)")
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--d", type=float, default=0.2)
    ap.add_argument("--seed", type=int, default=20240601)
    ap.add_argument("--cpu-rows", type=int, default=1_000_000,
                    help="rows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--config", default="dd",
                    choices=["dd", "gp_nich", "dpd", "bb", "gp", "nich", "mixed"],
                    help="dd = the headline workload (BASELINE configs[1]); "
                         "the others are the remaining BASELINE configs, for "
                         "DESIGN.md's table (not the bench line of record)")
    ap.add_argument("--value-sorted", type=int, default=1,
                    help="0 generic kernel only, 1 auto, 2 force")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="keep the per-batch all-reduce on torch.distributed "
                         "instead of the library's own RCCL communicator")
    ap.add_argument("--force-collective", action="store_true",
                    help="diagnostic: run the N>1 code path (statistic "
                         "deltas + RCCL all-reduce) with a single rank")
    return ap.parse_args()


def measured_traffic(kernel, rows_per_launch):
    """HBM bytes per launch of `kernel` from the committed PMC passes
    (profiles/r1_traffic.json: FETCH_SIZE + WRITE_SIZE, separate rocprofv3
    --pmc runs of this same command), scaled to this run's rows per launch.
    None when no measurement is on file for the kernel."""
    path = os.path.join(ROOT, "profiles", "r1_traffic.json")
    try:
        rec = json.load(open(path))[kernel]
    except (OSError, KeyError, ValueError):
        return None
    kb = rec["fetch_kb_per_launch"] + rec["write_kb_per_launch"]
    return kb * 1024.0 * rows_per_launch / rec["rows_per_launch"]


def cpu_baseline(args):
    """Oracle (port of the reference loop) on a bounded sample.  The figure of
    record is ONE thread (the reference is single-threaded); `all_cores` adds
    what the host reaches with one independent chain per core."""
    import threading
    import numpy as np
    import oracle_lib as ol

    def chain(n, seed):
        rng = np.random.default_rng(seed)
        values = rng.integers(0, args.dim, n).astype(np.uint32)
        assign = (np.arange(n) % args.groups).astype(np.uint32)
        orc = ol.OracleMixture(args.alpha, args.d, [
            ol.make_shared(ol.DD, alphas=[0.5] * args.dim)])
        orc.init_from_assignments([values], assign, args.groups, 1)
        return orc

    n = min(args.cpu_rows, args.rows)
    orc = chain(n, args.seed)
    st = ol.oracle().orc_rng_seed(args.seed)
    t0 = time.perf_counter()
    orc.gibbs_sequential(0, n, st)
    dt = time.perf_counter() - t0

    cores = os.cpu_count() or 1
    n_par = max(10000, n // 8)
    chains = [chain(n_par, args.seed + 1 + i) for i in range(cores)]
    threads = [threading.Thread(target=c.gibbs_sequential, args=(0, n_par, st))
               for c in chains]          # ctypes releases the GIL
    t1 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt_par = time.perf_counter() - t1
    return {
        "value": n / dt,
        "unit": "row-updates/s",
        "cores": 1,
        "kind": "port",
        "sample": "one sequential sweep over the first %d rows of the "
                  "workload (K=%d, dim=%d), oracle/oracle.c -O3, %.1f s"
                  % (n, args.groups, args.dim, dt),
        "all_cores": {"value": cores * n_par / dt_par, "cores": cores,
                      "sample": "%d independent chains of %d rows, %.1f s"
                                % (cores, n_par, dt_par)},
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    try:
        from distributions_amd import _core, engine
    except (ImportError, OSError):
        # a checkout without the built libraries: compile them, then go on
        # (there is no other way to run: the product has no CPU path)
        import __graft_entry__
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            __graft_entry__.build()
        else:
            time.sleep(240)
        from distributions_amd import _core, engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    _core.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n = args.rows
    k = args.groups
    row_offset = rank * n
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed + rank)
    assign = (torch.arange(n, device=dev, dtype=torch.int64)
              + row_offset).remainder(k).to(torch.int32)

    def poisson(mean):
        return torch.poisson(torch.full((n,), mean, device=dev),
                             generator=gen).to(torch.int32)

    def normal():
        return torch.randn((n,), generator=gen, device=dev,
                           dtype=torch.float32)

    if args.config == "dd":
        columns = [torch.randint(0, args.dim, (n,), generator=gen, device=dev,
                                 dtype=torch.int32)]
        shareds = [engine.dd_shared([0.5] * args.dim)]
        bytes_per_row = 12 * k + 12       # 4K (PY) + 2*4K (DD) + 12
    elif args.config == "dpd":
        columns = [torch.randint(0, args.dim, (n,), generator=gen, device=dev,
                                 dtype=torch.int32)]
        shareds = [engine.dpd_shared(0.5, [1.0 / args.dim] * args.dim, 0.0)]
        bytes_per_row = 12 * k + 12
    elif args.config == "bb":
        columns = [(torch.rand((n,), generator=gen, device=dev) < 0.3).to(
            torch.int32)]
        shareds = [engine.bb_shared(0.5, 2.0)]
        bytes_per_row = 8 * k + 12
    elif args.config == "gp":
        columns = [poisson(5.0)]
        shareds = [engine.gp_shared(1.0, 1.0)]
        bytes_per_row = 16 * k + 12
    elif args.config == "nich":
        columns = [normal()]
        shareds = [engine.nich_shared(0.0, 1.0, 1.0, 1.0)]
        bytes_per_row = 20 * k + 12
    elif args.config == "mixed":
        # a row of mixed type, the shape of real tables: two categoricals, a
        # boolean, a count and a real (run-time feature list in the kernel)
        columns = [torch.randint(0, 16, (n,), generator=gen, device=dev,
                                 dtype=torch.int32),
                   torch.randint(0, 4, (n,), generator=gen, device=dev,
                                 dtype=torch.int32),
                   (torch.rand((n,), generator=gen, device=dev) < 0.3).to(
                       torch.int32),
                   poisson(5.0), normal()]
        shareds = [engine.dd_shared([0.5] * 16), engine.dd_shared([0.5] * 4),
                   engine.bb_shared(0.5, 2.0), engine.gp_shared(1.0, 1.0),
                   engine.nich_shared(0.0, 1.0, 1.0, 1.0)]
        bytes_per_row = (1 + 2 + 2 + 1 + 3 + 4) * 4 * k + 28
    else:   # gp_nich: BASELINE configs[2]
        columns = [poisson(5.0), normal()]
        shareds = [engine.gp_shared(1.0, 1.0),
                   engine.nich_shared(0.0, 1.0, 1.0, 1.0)]
        bytes_per_row = 32 * k + 16       # (1+3+4)*4K + 16
    g = engine.Gibbs(args.alpha, args.d, shareds)
    g.set_option("value_sorted", args.value_sorted)
    initial = assign.clone()   # the engine keeps updating `assign` in place
    g.load_rows_torch(columns, assign, k, 1, row_offset=row_offset)
    sharded = engine.ShardedGibbs(g.core, n, row_offset, device=dev,
                                  force_collective=args.force_collective,
                                  columns=columns, assign_packed=initial)
    sharded.sync_initial_stats()
    native_comm = (not args.torch_collectives) and sharded.use_native_comm()
    seed_state = _core.rng_seed(args.seed)

    def step(i):
        # every sweep draws a fresh stretch of the engine's stream
        sharded.sweep(args.batch, seed_state, draw_base=i * n * world)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    g.kernel_stats(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms, launches, rows = g.kernel_stats()
    total_rows = float(n) * world * args.steps
    vs_batches, generic_batches = g.path_counts()
    kernel = ("k_vs_sample" if vs_batches else "k_sweep_sample") + "<%s>" % (
        args.config)
    if rank == 0:
        out = {
            "metric": "row-Gibbs-updates/sec (score+sample+suffstat) at "
                      "N=10M, K=1024; 1/2/4/8 GPU",
            "value": total_rows / dt,
            "unit": "row-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s N=%d rows/GPU K=%d+1 "
                            "PitmanYor(alpha=%g,d=%g), frozen sub-sweeps of "
                            "%d rows" % (
                                {"dd": "DirichletDiscrete(dim=%d)" % args.dim,
                                 "dpd": "DirichletProcessDiscrete(V=%d)"
                                        % args.dim,
                                 "bb": "BetaBernoulli", "gp": "GammaPoisson",
                                 "nich": "NormalInverseChiSq",
                                 "gp_nich": "GammaPoisson+NormalInverseChiSq",
                                 "mixed": "DD(16)+DD(4)+BetaBernoulli+"
                                          "GammaPoisson+NormalInverseChiSq",
                                 }[args.config],
                                n, k, args.alpha, args.d, args.batch),
                "rows_per_gpu": n, "groups": k, "dim": args.dim,
                "batch_rows": args.batch,
                "parallelism": "rows sharded over %d GPU(s), all-reduce of "
                               "statistic deltas per sub-sweep" % world,
                "collectives": ("none" if not sharded.collective else
                                "library RCCL communicator" if native_comm
                                else "torch.distributed (RCCL)"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel,
                "achieved": (bytes_per_row * rows / max(launches, 1))
                            / (1e-3 * ms / max(launches, 1)) / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (bytes_per_row * rows) / (1e-3 * ms) / 1e9
                        / HBM_PEAK_GBS,
                "traffic": measured_traffic(kernel, rows / max(launches, 1)),
                "algorithmic_bytes_per_row": bytes_per_row,
                "rows_per_launch": rows / max(launches, 1),
                "avg_launch_ms": ms / max(launches, 1),
                "launches": launches,
                "note": "achieved = SURVEY 8d algorithmic bytes / kernel time; "
                        "the value-sorted kernels serve those bytes from "
                        "scalar-loaded per-value tables (HBM carries `traffic`) "
                        "and are VALU-issue-bound: see DESIGN.md section 4",
            },
        }
        if world == 1 and args.cpu_rows > 0 and args.config == "dd":
            out["cpu_baseline"] = cpu_baseline(args)
        # RCCL prints its version banner through C stdio; push that out first
        # so that the JSON line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
