#!/usr/bin/env python3
"""bench.py -- row-Gibbs-updates/s of the batched mixture row update.

Workload (BASELINE.json configs[1], SURVEY 8d "C2"): DirichletDiscrete
dim=256, alphas=0.5, K=1024 non-empty groups + 1 empty, N=10M rows per GPU,
values iid uniform{0..255}, initial assignment i mod K, PitmanYor(alpha=1,
d=0.2).  A step = one full Gibbs pass over the resident rows (self-remove,
score all K groups, sample, add), in frozen sub-sweeps of --batch rows, with
the statistics update, group-set normalisation and cache rebuild included.
Rows are generated on the device (seeded) before the timed region.

`--gpus N` with N > 1 starts N ranks itself (one process per GPU, backend
nccl == RCCL) unless a launcher (torch.distributed.run) already set
WORLD_SIZE; the parent never touches a GPU.  Rows are sharded, weak scaling
(N rows per GPU), one all-reduce of the integer statistic deltas per
sub-sweep; `strong_scaling` adds the same job with --rows split over the
ranks.

Prints ONE JSON line (rank 0).  `roofline` prices the score+sample kernel
(k_vs_sample, or k_sweep_sample / k_sweep_program where the value-sorted kernel
does not apply): its HIP-event duration on the launch stream is measured in
this run; the counters behind `frac` (VALU busy cycles for the VALU-bound
kernels, HBM bytes for the HBM-bound one) come from the rocprofv3 --pmc passes
of this same command committed under profiles/ (profiles/r3_counters.json,
ignored when the kernel sources changed since).  `cpu_baseline` times the
oracle's sequential chain (the reference loop restated, oracle/oracle.c) on the
first --cpu-rows rows of the very column the GPU holds, one host thread.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
SIMDS = 1024            # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4         # MI355X_MICROARCH.md: max clock
COUNTERS_FILE = "r6_counters.json"   # tools/counters.py, this round's sources
# what one wave64 vector instruction costs its SIMD when issued back to back,
# by class (tools/microbench/valu_issue.hip, profiles/r2_final_valu_issue.txt)
ISSUE_CYCLES_PACKED = 4.19   # v_pk_*, also shifts and compares
ISSUE_CYCLES_PLAIN = 2.3     # plain f32 / int32 operations
METRIC = ("row-Gibbs-updates/sec (score+sample+suffstat) at N=10M, K=1024; "
          "1/2/4/8 GPU")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
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
                    choices=["dd", "dd16", "gp_nich", "dpd", "bb", "gp",
                             "nich", "mixed"],
                    help="dd = the headline workload (BASELINE configs[1]); "
                         "the others are the remaining BASELINE configs, for "
                         "DESIGN.md's table (not the bench line of record)")
    ap.add_argument("--values", default="uniform",
                    choices=["uniform", "zipf"],
                    help="categorical columns: iid uniform, or Zipf(1.1) "
                         "(SURVEY 8d's skewed variant of C2)")
    ap.add_argument("--value-sorted", type=int, default=1,
                    help="0 generic kernel only, 1 auto, 2 force")
    ap.add_argument("--value-stream", type=int, default=1,
                    help="the table-free value-sorted kernel: 0 never, "
                         "1 auto, 2 always")
    ap.add_argument("--device-normalise", type=int, default=2,
                    help="the group set is normalised on the device, no host "
                         "round trip per sub-sweep: 0 never, 1 or 2 (the "
                         "library's default) where it applies")
    ap.add_argument("--narrow-tiles", type=int, default=1,
                    help="launches too small to fill the chip take tiles of "
                         "64 rows with their vectors in LDS (k_vs_narrow): "
                         "0 never, 1 auto, 2 whenever the vectors fit")
    ap.add_argument("--stream-scratch", type=int, default=1,
                    help="k_vs_stream keeps the first pass's likelihoods for "
                         "the scan in a scratch row per tile: 1 or 0")
    ap.add_argument("--kernel-timing", type=int, default=8,
                    help="HIP events around every n-th score+sample launch "
                         "of the timed region (the roofline's duration is "
                         "their average; two events cost a sub-sweep 8 us): "
                         "1 all, 0 none")
    ap.add_argument("--other-batches", default="65536",
                    help="comma-separated sub-sweep sizes timed besides "
                         "--batch (a few steps each, reported in "
                         "`batch_variants`; empty = none)")
    ap.add_argument("--other-configs", default="gp_nich,mixed,dpd",
                    help="N = 1: the other BASELINE configurations timed "
                         "besides the headline one (a few steps each, exact "
                         "and scan sampling, reported in `other_configs` and "
                         "as scalars in `config`; dpd = BASELINE configs[4] "
                         "at K = 8192, V = 10 000; empty = none)")
    ap.add_argument("--opt", action="append", default=[],
                    metavar="NAME=VALUE",
                    help="any other engine option (dist_gibbs_set_option), "
                         "e.g. rows_scratch=0; repeatable")
    ap.add_argument("--no-breakdown", action="store_true",
                    help="skip the diagnostic pass behind `step_breakdown` "
                         "(two sweeps with events at the phase boundaries)")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="keep the per-batch all-reduce on torch.distributed "
                         "instead of the library's own RCCL communicator")
    ap.add_argument("--force-collective", action="store_true",
                    help="diagnostic: run the N>1 code path (statistic "
                         "deltas + RCCL all-reduce) with a single rank")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1: skip the strong-scaling leg")
    ap.add_argument("--placement", default="auto",
                    choices=["auto", "block", "value"],
                    help="N > 1: how rows are placed on the ranks.  value = "
                         "a rank holds the rows of ITS range of values of "
                         "the (one, categorical) feature, so the cells never "
                         "travel and a sub-sweep exchanges 3 words per group "
                         "(dist_gibbs_partition_by_value); block = any rows "
                         "anywhere, 3 + dim words per group; auto = value "
                         "where the feature list allows it")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="N = 1: after the timed region the same job runs on "
                         "for at least this long (`sustained_value`: the "
                         "group count keeps creeping up; 0 = skip)")
    ap.add_argument("--exact-chains", type=int, default=1024,
                    help="N = 1: independent exact chains (the reference's "
                         "own sampler) run concurrently for "
                         "`exact_chains_value`; 0 = skip")
    ap.add_argument("--exact-rows", type=int, default=20000,
                    help="rows each exact chain walks in its timed call "
                         "(a call costs the host some 26 us per engine -- "
                         "mirrors pulled, buffers reserved -- beside 9-20 us "
                         "per row on the device)")
    return ap.parse_args()


# ---------------------------------------------------------------------------
# the launcher: `python bench.py --gpus N` starts its own ranks

def libraries_built():
    import glob
    return (os.path.exists(os.path.join(ROOT, "distributions_amd",
                                        "libdistributions_hip.so"))
            and glob.glob(os.path.join(ROOT, "distributions_amd", "_core*.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")))


def ensure_built():
    """Compile what a bare checkout lacks, in a child process (the product has
    no CPU path to fall back on, and this process must stay off the GPU)."""
    if libraries_built():
        return
    import fcntl
    # ranks of a launcher start together: one builds, the others wait here
    with open(os.path.join(ROOT, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not libraries_built():
                subprocess.check_call(
                    [sys.executable, "-c",
                     "import __graft_entry__; __graft_entry__.build()"],
                    cwd=ROOT)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def launch_ranks(args):
    """Parent of an N-rank run: builds once, checks the device count, starts
    one child per GPU and exits with their status.  Touches no GPU itself
    (torch.cuda.device_count() does not initialise one on this image)."""
    ensure_built()
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write(
            "bench.py: --gpus %d asked for, %d GPU(s) visible: refusing to "
            "report an N-GPU number from fewer devices\n" % (args.gpus, have))
        return 2
    with socket.socket() as s:    # a free rendezvous port for this run
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
            env=env, stdout=None if rank == 0 else subprocess.DEVNULL))
    status = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and status == 0:
                status = rc
                for q in live:     # a rank died: its peers would hang
                    q.terminate()
        time.sleep(0.05)
    return status


# ---------------------------------------------------------------------------

def source_hash():
    """What the committed counters were measured on: the kernel sources."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "distributions_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) and name != "ref_tables.h":
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def committed_counters(kernel):
    """Per-launch counter means of `kernel` from the committed rocprofv3
    --pmc passes (profiles/r5_counters.json, written by tools/counters.py on
    the GPU box from separate passes of this command).  None when there is no
    record or the kernel sources changed since it was taken."""
    path = os.path.join(ROOT, "profiles", COUNTERS_FILE)
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    k = rec.get("kernels", {}).get(kernel)
    if k is None:
        return None
    k = dict(k)
    k["stale"] = rec.get("source_hash") != source_hash()
    return k


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def cpu_baseline(args, column_host):
    """Oracle (port of the reference loop) on a bounded sample: the first
    `cpu_rows` rows of the column the GPU holds.  The figure of record is ONE
    thread (the reference is single-threaded); `all_cores` adds what the host
    reaches with one independent chain per core, `march_native` the same
    single chain from a -march=native build of the same file."""
    import threading
    import numpy as np
    import oracle_lib as ol

    dim = 16 if args.config == "dd16" else args.dim

    def chain(values, lib=None):
        n = len(values)
        assign = (np.arange(n) % args.groups).astype(np.uint32)
        orc = ol.OracleMixture(args.alpha, args.d, [
            ol.make_shared(ol.DD, alphas=[0.5] * dim)], lib=lib)
        orc.init_from_assignments([values], assign, args.groups, 1)
        return orc

    n = len(column_host)
    if n < 8 * args.groups:
        # (every group of the initial assignment i mod K needs rows)
        raise SystemExit("bench.py: --cpu-rows must be at least 8 x --groups")
    values = np.ascontiguousarray(column_host, np.uint32)
    orc = chain(values)
    st = ol.oracle().orc_rng_seed(args.seed)
    t0 = time.perf_counter()
    orc.gibbs_sequential(0, n, st)
    dt = time.perf_counter() - t0

    model, cores = host_cpu()
    n_par = min(n, max(10 * args.groups, n // 8))
    chains = [chain(np.roll(values, 7919 * (i + 1))[:n_par].copy())
              for i in range(cores)]
    threads = [threading.Thread(target=c.gibbs_sequential, args=(0, n_par, st))
               for c in chains]          # ctypes releases the GIL
    t1 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt_par = time.perf_counter() - t1
    out = {
        "value": n / dt,
        "unit": "row-updates/s",
        "cores": 1,
        "kind": "port",
        "cpu": model,
        "host_cores": cores,
        "sample": "one sequential sweep (the reference's loop, "
                  "examples/mixture/main.py:236-244) over the first %d rows "
                  "of the GPU's own column (copied to the host), K=%d, "
                  "dim=%d; oracle/oracle.c, gcc -O3 -msse4.1 -fno-fast-math, "
                  "%.1f s" % (n, args.groups, dim, dt),
        "all_cores": {"value": cores * n_par / dt_par, "cores": cores,
                      "sample": "%d independent chains of %d rows, %.1f s"
                                % (cores, n_par, dt_par)},
    }
    out["c1"] = cpu_c1(args, ol)
    out["reference_kernels"] = cpu_reference_kernels()
    # the same file built for this host's own instruction set
    try:
        subprocess.check_call(["make", "-s", "-C",
                               os.path.join(ROOT, "oracle"), "native"],
                              stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        native = ol.bind_oracle(os.path.join(ROOT, "oracle", "_native",
                                             "liboracle_native.so"))
        n_nat = min(n, 200_000)
        orc2 = chain(values[:n_nat].copy(), native)
        t2 = time.perf_counter()
        orc2.gibbs_sequential(0, n_nat, st)
        dt2 = time.perf_counter() - t2
        out["march_native"] = {
            "value": n_nat / dt2, "cores": 1,
            "sample": "first %d rows, gcc -O3 -march=native "
                      "-fno-fast-math -ffp-contract=off, %.1f s"
                      % (n_nat, dt2)}
    except (subprocess.CalledProcessError, OSError, AttributeError) as e:
        out["march_native"] = {"value": None, "error": str(e)[:200]}
    return out


def cpu_reference_kernels():
    """The REAL reference where it compiles (oracle/_ref/libref.so: its
    src/vector_math.cc and src/special.cc built with its release flags, see
    oracle/Makefile): the three streaming loops a row update spends its time
    in -- vector_add_subtract (vector_math.cc:160-168, the DirichletDiscrete
    score), vector_max (:74-83) and vector_exp (:190-221, fmath's table exp)
    -- at K = 1024 and K = 64, one thread.  None when the library is absent."""
    import ctypes
    import numpy as np
    path = os.path.join(ROOT, "oracle", "_ref", "libref.so")
    try:
        L = ctypes.CDLL(path)
    except OSError:
        return None
    fp = ctypes.POINTER(ctypes.c_float)
    L.ref_vector_add_subtract.argtypes = [ctypes.c_size_t, fp, fp, fp]
    L.ref_vector_max.argtypes = [ctypes.c_size_t, fp]
    L.ref_vector_max.restype = ctypes.c_float
    L.ref_vector_exp.argtypes = [ctypes.c_size_t, fp]
    out = {}
    rng = np.random.default_rng(1)
    for k in (1024, 64):
        a = rng.normal(size=k).astype(np.float32)
        b = rng.normal(size=k).astype(np.float32)
        io = np.zeros(k, np.float32)
        pa, pb, pio = (x.ctypes.data_as(fp) for x in (a, b, io))
        reps = 20000 if k == 1024 else 200000

        def rate(fn):
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            return k * reps / (time.perf_counter() - t0) / 1e6

        def exp_once():
            io[:] = a            # (keeps the argument in fmath's range)
            L.ref_vector_exp(k, pio)
        out["K=%d" % k] = {
            "vector_add_subtract_elements_per_us":
                rate(lambda: L.ref_vector_add_subtract(k, pio, pa, pb)),
            "vector_max_elements_per_us":
                rate(lambda: L.ref_vector_max(k, pa)),
            "vector_exp_elements_per_us": rate(exp_once),
        }
    out["note"] = ("the reference's own vector_math.cc / special.cc "
                   "(-O3 -msse4.1 -ffast-math), called through ctypes: the "
                   "call overhead (about 1 us) is inside the K = 64 figures")
    return out


def cpu_c1(args, ol):
    """BASELINE configs[0] on the host, next to the survey's probes of the
    compiled reference: the full row update at
    DirichletDiscrete(16), K = 64, N = 100 000 (probe: 1.83 M rows/s) and the
    loop of benchmarks/mixture.cc:104-115 -- remove, score_value, add; its
    "cells/us" -- for DirichletDiscrete<4> at K = 1000 (probe: 9.43)."""
    import numpy as np
    rng = np.random.default_rng(args.seed)
    n, k, dim = 100_000, 64, 16
    values = rng.integers(0, dim, n).astype(np.uint32)
    orc = ol.OracleMixture(1.0, 0.0, [ol.make_shared(ol.DD,
                                                     alphas=[0.5] * dim)])
    orc.init_from_assignments([values], (np.arange(n) % k).astype(np.uint32),
                              k, 1)
    st = ol.oracle().orc_rng_seed(args.seed)
    orc.gibbs_sequential(0, n, st)            # warm-up sweep
    t0 = time.perf_counter()
    orc.gibbs_sequential(0, n, st)
    full = n / (time.perf_counter() - t0)
    k2, n2, iters = 1000, 4000, 400_000
    v2 = rng.integers(0, 4, n2).astype(np.uint32)
    g2 = rng.integers(0, k2, n2).astype(np.uint32)
    m = ol.OracleMixture(1.0, 0.0, [ol.make_shared(ol.DD, alphas=[0.5] * 4)])
    m.init_from_assignments([v2], g2, k2, 1)
    t0 = time.perf_counter()
    m.L.orc_mixture_benchmark_loop(m.h, n2, v2, g2, iters)
    loop = iters / (time.perf_counter() - t0) / 1e6
    return {"full_row_update_dd16_k64_rows_per_s": full,
            "reference_probe_rows_per_s": 1.83e6,
            "mixture_cc_loop_dd4_k1000_cells_per_us": loop,
            "mixture_cc_loop_label": "UNVECTORISED port (oracle.c is built "
                                     "-fno-fast-math, its loop stays "
                                     "scalar); the compiled reference's "
                                     "probe below is the figure to compare "
                                     "a GPU number with",
            "reference_probe_cells_per_us": 9.43,
            "note": "oracle/oracle.c on one host core; the probes are the "
                    "compiled reference in the survey container (another "
                    "CPU), BASELINE.md section 2"}


def make_columns(args, torch, engine, dev, gen, n, k, value_range=None):
    """value_range = (rank, world): the categorical column takes values of
    this rank's share of the domain only (value-partitioned placement)"""
    def poisson(mean):
        return torch.poisson(torch.full((n,), mean, device=dev),
                             generator=gen).to(torch.int32)

    def normal():
        return torch.randn((n,), generator=gen, device=dev,
                           dtype=torch.float32)

    def categorical(dim):
        lo, hi = 0, dim
        if value_range is not None:
            r, w_ = value_range
            lo, hi = r * dim // w_, (r + 1) * dim // w_
        if args.values == "zipf":
            # Zipf(s = 1.1) over the dim values, by inversion
            # (a rank's share of the domain: the law restricted to it)
            w = 1.0 / torch.arange(lo + 1, hi + 1, device=dev,
                                   dtype=torch.float64) ** 1.1
            cdf = torch.cumsum(w / w.sum(), 0).to(torch.float32)
            u = torch.rand((n,), generator=gen, device=dev)
            return (torch.searchsorted(cdf, u).clamp_(max=hi - lo - 1)
                    + lo).to(torch.int32)
        return torch.randint(lo, hi, (n,), generator=gen, device=dev,
                             dtype=torch.int32)

    if args.config in ("dd", "dd16"):
        dim = 16 if args.config == "dd16" else args.dim
        columns = [categorical(dim)]
        shareds = [engine.dd_shared([0.5] * dim)]
        bytes_per_row = 12 * k + 12       # 4K (PY) + 2*4K (DD) + 12
        name = "DirichletDiscrete(dim=%d)" % dim
    elif args.config == "dpd":
        columns = [categorical(args.dim)]
        shareds = [engine.dpd_shared(0.5, [1.0 / args.dim] * args.dim, 0.0)]
        bytes_per_row = 12 * k + 12
        name = "DirichletProcessDiscrete(V=%d)" % args.dim
    elif args.config == "bb":
        columns = [(torch.rand((n,), generator=gen, device=dev) < 0.3).to(
            torch.int32)]
        shareds = [engine.bb_shared(0.5, 2.0)]
        bytes_per_row = 8 * k + 12
        name = "BetaBernoulli"
    elif args.config == "gp":
        columns = [poisson(5.0)]
        shareds = [engine.gp_shared(1.0, 1.0)]
        bytes_per_row = 16 * k + 12
        name = "GammaPoisson"
    elif args.config == "nich":
        columns = [normal()]
        shareds = [engine.nich_shared(0.0, 1.0, 1.0, 1.0)]
        bytes_per_row = 20 * k + 12
        name = "NormalInverseChiSq"
    elif args.config == "mixed":
        # a row of mixed type, the shape of real tables: two categoricals, a
        # boolean, a count and a real (run-time feature list in the kernel)
        columns = [categorical(16), categorical(4),
                   (torch.rand((n,), generator=gen, device=dev) < 0.3).to(
                       torch.int32),
                   poisson(5.0), normal()]
        shareds = [engine.dd_shared([0.5] * 16), engine.dd_shared([0.5] * 4),
                   engine.bb_shared(0.5, 2.0), engine.gp_shared(1.0, 1.0),
                   engine.nich_shared(0.0, 1.0, 1.0, 1.0)]
        bytes_per_row = (1 + 2 + 2 + 1 + 3 + 4) * 4 * k + 28
        name = "DD(16)+DD(4)+BetaBernoulli+GammaPoisson+NormalInverseChiSq"
    else:   # gp_nich: BASELINE configs[2]
        columns = [poisson(5.0), normal()]
        shareds = [engine.gp_shared(1.0, 1.0),
                   engine.nich_shared(0.0, 1.0, 1.0, 1.0)]
        bytes_per_row = 32 * k + 16       # (1+3+4)*4K + 16
        name = "GammaPoisson+NormalInverseChiSq"
    return columns, shareds, bytes_per_row, name


def run_rank(args):
    import torch
    import torch.distributed as dist
    ensure_built()    # (under a launcher rank 0 of a bare checkout builds;
    #                   launch_ranks has done it already for our own children)
    from distributions_amd import _core, engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n"
                         % (args.gpus, world))
        return 2
    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d has no GPU %d (%d visible)\n"
                         % (rank, local_rank, torch.cuda.device_count()))
        return 2
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    _core.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    k = args.groups
    seed_state = _core.rng_seed(args.seed)

    by_value = (sharded_collective(world, args)
                and not args.torch_collectives
                and args.config in ("dd", "dd16", "dpd")
                and args.placement in ("auto", "value"))
    if args.placement == "value" and not by_value:
        sys.stderr.write("bench.py: --placement value needs N > 1 (or "
                         "--force-collective), the library's communicator "
                         "and one categorical feature\n")
        return 2

    def build_job(n, row_offset, args=args):
        k = args.groups   # (a sub-job may have its own: C5 runs at K = 8192)
        gen = torch.Generator(device=dev)
        gen.manual_seed(args.seed + rank)
        assign = (torch.arange(n, device=dev, dtype=torch.int64)
                  + row_offset).remainder(k).to(torch.int32)
        columns, shareds, bytes_per_row, name = make_columns(
            args, torch, engine, dev, gen, n, k,
            (rank, world) if by_value else None)
        g = engine.Gibbs(args.alpha, args.d, shareds)
        g.set_option("value_sorted", args.value_sorted)
        g.set_option("value_stream", args.value_stream)
        g.set_option("device_normalise", args.device_normalise)
        g.set_option("narrow_tiles", args.narrow_tiles)
        g.set_option("kernel_timing", args.kernel_timing)
        g.set_option("debug.stream_scratch", args.stream_scratch)
        for item in args.opt:
            key, _, val = item.partition("=")
            g.set_option(key.strip(), int(val))
        initial = assign.clone()   # the engine updates `assign` in place
        g.load_rows_torch(columns, assign, k, 1, row_offset=row_offset)
        sharded = engine.ShardedGibbs(
            g.core, n, row_offset, device=dev,
            force_collective=args.force_collective, columns=columns,
            assign_packed=initial)
        sharded.sync_initial_stats()
        return g, sharded, columns, bytes_per_row, name

    def timed(sharded, g, n, batch, steps, warmup, first_draw):
        """-> seconds for `steps` sweeps (max over ranks)"""
        def step(i):
            # every sweep draws a fresh stretch of the engine's stream
            sharded.sweep(batch, seed_state,
                          draw_base=(first_draw + i) * n * world)
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        g.kernel_stats(reset=True)
        g.core.comm_volume(reset=True)   # (counters only: the run stays open)
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        timed.host_enqueue_s = time.perf_counter() - t0
        # the host's copy of the group set is part of the job: a sweep may
        # leave it to be pulled on demand (device-normalised runs stay open),
        # so it is demanded here, inside the timed region
        len(g)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    n = args.rows
    g, sharded, columns, bytes_per_row, name = build_job(n, rank * n)
    native_comm = (not args.torch_collectives) and sharded.use_native_comm()
    if by_value:
        if not native_comm:
            sys.stderr.write("bench.py: value placement without the "
                             "library's communicator\n")
            return 2
        sharded.partition_by_value()
    column_host = None
    if (rank == 0 and world == 1 and args.cpu_rows > 0
            and args.config in ("dd", "dd16")):
        column_host = columns[0][:min(args.cpu_rows, n)].cpu().numpy()

    dt = timed(sharded, g, n, args.batch, args.steps, args.warmup, 0)
    host_enqueue_ms = 1e3 * timed.host_enqueue_s / max(args.steps, 1)
    ms, launches, rows = g.kernel_stats()
    vs_batches, generic_batches = g.path_counts()
    # what the engine that ran the timed region says about itself (read
    # here: `g` is rebuilt further down)
    timed_counts = g.core.debug_counts()
    # (what the exchanges of the timed region really sent: the counters of
    # dist_gibbs_sweep_sharded, not the size of a closed run's image)
    volume = g.core.comm_volume() if native_comm else None
    streamed = timed_counts["stream_batches"]
    narrow = timed_counts["narrow_batches"]
    scratched = timed_counts["scratch_batches"]
    scanned = timed_counts["scan_batches"]
    draws = args.warmup + args.steps
    groups_at_end = len(g)
    comm_ms, comm_count = (g.core.comm_stats() if native_comm else (0.0, 0))

    # where a sub-sweep's time goes: two more sweeps with events at the phase
    # boundaries (a diagnostic pass of its own: the events cost some 20 us a
    # sub-sweep, so the line's `value` is not measured with them)
    breakdown = None
    if (rank == 0 and world == 1 and vs_batches and not args.force_collective
            and not args.no_breakdown):
        g.set_option("phase_timing", 1)
        g.core.phase_stats(reset=True)
        timed(sharded, g, n, args.batch, 2, 0, draws)
        draws += 2
        ms5, timed_batches = g.core.phase_stats(reset=True)
        g.set_option("phase_timing", 0)
        if timed_batches:
            names = ("tables_per_value", "score_and_sample",
                     "handed_over_rows", "statistics",
                     "group_set_and_caches")
            breakdown = {name: 1e3 * ms / timed_batches
                         for name, ms in zip(names, ms5)}
            breakdown["unit"] = "us per sub-sweep (HIP events)"
            breakdown["sub_sweeps_timed"] = timed_batches


    # the same job at other sub-sweep sizes (value depends on it: the
    # per-batch kernels and launch gaps do not shrink with the batch)
    variants = []
    for b in [int(x) for x in args.other_batches.split(",") if x.strip()]:
        if b == args.batch or b <= 0:
            continue
        steps_b = max(1, min(args.steps, 5))
        before = g.core.debug_counts()
        groups_before = len(g)
        dt_b = timed(sharded, g, n, b, steps_b, 1, draws)
        draws += 1 + steps_b
        ms_b, launches_b, rows_b = g.kernel_stats()
        after = g.core.debug_counts()
        took = [name for key, name in (("scratch_batches", "k_rows_scratch"),
                                       ("narrow_batches", "k_vs_narrow"),
                                       ("stream_batches", "k_vs_stream"),
                                       ("value_sorted_batches", "k_vs_sample"))
                if after[key] > before[key]]
        rec = {
            "batch_rows": b, "value": float(n) * world * steps_b / dt_b,
            "ms_per_step": 1e3 * dt_b / steps_b, "steps": steps_b,
            "host_enqueue_ms_per_step": 1e3 * timed.host_enqueue_s / steps_b,
            "kernel": took[0] if took else "k_sweep_sample",
            "kernel_avg_launch_ms": ms_b / max(launches_b, 1),
            # (the chain goes on from the timed region: smaller sub-sweeps
            # found groups faster -- every one of them may fill the empty
            # group -- and every kernel's work follows the group count)
            "groups_at_start": groups_before, "groups_at_end": len(g)}
        if world == 1 and not args.force_collective:
            # ... and the same sub-sweep size on a chain of its own, from the
            # initial assignment, with the headline's warm-up
            g3, sh3, _, _, _ = build_job(n, 0)
            dt_f = timed(sh3, g3, n, b, steps_b, args.warmup, 0)
            rec["fresh_chain_value"] = float(n) * steps_b / dt_f
            rec["fresh_chain_groups_at_end"] = len(g3)
            del g3, sh3
            torch.cuda.empty_cache()
        variants.append(rec)

    # (after the variants: they are quoted at the group count the timed region
    # left)
    # the same job running on (N = 1): the line's `value` is 20 sweeps, 22 ms
    # of device time; this is the rate over seconds of them, with the group
    # count the chain has reached by then (every kernel's work follows it)
    sustained = None
    if (world == 1 and not args.force_collective
            and args.sustained_seconds > 0 and dt > 0):
        # (in stretches of what the timed region's rate predicts, until the
        # clock says so: the estimate is off both ways -- a long run pulls
        # the host's copy of the group set once, and slows as groups are
        # founded)
        chunk = max(args.steps,
                    int(args.sustained_seconds / (dt / args.steps)) + 1)
        steps_s, dt_s = 0, 0.0
        while dt_s < args.sustained_seconds:
            dt_s += timed(sharded, g, n, args.batch, chunk, 0, draws)
            draws += chunk
            steps_s += chunk
            chunk = max(args.steps, chunk // 4)
        sustained = {"value": float(n) * steps_s / dt_s,
                     "unit": "row-updates/s", "steps": steps_s,
                     "seconds": dt_s, "ms_per_step": 1e3 * dt_s / steps_s,
                     "groups_at_end": len(g)}

    # the general-row configurations (any feature list the value-sorted
    # kernels do not take), exact and with scan sampling
    others = []
    if world == 1 and not args.force_collective:
        for cfg in [c.strip() for c in args.other_configs.split(",")
                    if c.strip()]:
            if cfg == args.config:
                continue
            for label, opts in (
                    ("exact", []),
                    ("scan sampling (tolerance-level, opt-in)",
                     ["sampling=1"]),
                    ("scan sampling + merged float statistics "
                     "(tolerance-level, opt-in)",
                     ["sampling=1", "float_stats=1"])):
                if cfg == "dpd" and "float_stats=1" in opts:
                    continue   # (integer statistics only)
                sub = argparse.Namespace(**vars(args))
                sub.config = cfg
                sub.opt = list(args.opt) + opts
                if cfg == "dpd":   # BASELINE configs[4]
                    sub.groups, sub.dim = 8192, 10000
                if len(others) == 0:
                    del sharded, g, columns
                    g = sharded = columns = None
                torch.cuda.empty_cache()
                g2, sh2, _, _, name2 = build_job(n, 0, sub)
                steps_o = max(1, min(args.steps, 3))
                dt_o = timed(sh2, g2, n, args.batch, steps_o, 1, 0)
                ms_o, launches_o, _ = g2.kernel_stats()
                counts_o = g2.core.debug_counts()
                others.append({
                    "config": cfg, "workload": name2,
                    "sampling": label,
                    "value": float(n) * steps_o / dt_o,
                    "unit": "row-updates/s", "steps": steps_o,
                    "ms_per_step": 1e3 * dt_o / steps_o,
                    "batch_rows": args.batch,
                    "groups": sub.groups,
                    "kernel": ("k_vs_scan_rows" if counts_o["scan_batches"]
                               and counts_o["value_sorted_batches"]
                               else "k_vs_stream" if counts_o["stream_batches"]
                               else "k_vs_sample"
                               if counts_o["value_sorted_batches"]
                               else "k_rows_scratch, scan mode"
                               if "sampling=1" in opts
                               and counts_o["scratch_batches"]
                               else "k_rows_scratch"
                               if counts_o["scratch_batches"]
                               else "k_sweep_sample"),
                    "folded": bool(counts_o["fold_batches"]),
                    "kernel_avg_launch_ms": ms_o / max(launches_o, 1)})
                del g2, sh2
        if others:
            torch.cuda.empty_cache()
            g, sharded, columns, _, _ = build_job(n, rank * n)

    # the reference's OWN sampler (one row at a time, examples/mixture/main.py:
    # 236-244) on the device: one chain, and --exact-chains independent ones in
    # one launch (BASELINE configs[3]: "independent chains"), same model and
    # group count as the headline
    exact = None
    if (not args.force_collective and args.exact_chains > 0
            and args.config in ("dd", "dd16")):
        # (N > 1: every rank runs its own chains -- "independent chains
        # sharded 1/GPU", no exchange at all: replicas only -- and the line
        # carries the sum over ranks, timed between two barriers)
        m = args.exact_chains
        rows_c = args.exact_rows
        n_c = max(3 * rows_c + 200, 8 * k)
        dim_c = 16 if args.config == "dd16" else args.dim
        chains = []
        for i in range(m):
            gen = torch.Generator(device=dev)
            gen.manual_seed(args.seed + 1000 + i + 100000 * rank)
            col = torch.randint(0, dim_c, (n_c,), generator=gen, device=dev,
                                dtype=torch.int32)
            asg = torch.arange(n_c, device=dev,
                               dtype=torch.int64).remainder(k).to(torch.int32)
            c = engine.Gibbs(args.alpha, args.d,
                             [engine.dd_shared([0.5] * dim_c)])
            c.load_rows_torch([col], asg, k, 1)
            chains.append(c)
        cores_ = [c.core for c in chains]
        import numpy as np
        st = np.array([_core.rng_seed(args.seed + 7 * i + 70001 * rank)
                       for i in range(m)], np.uint32)

        def between_barriers(fn):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return out, dt
        st1 = chains[0].sweep_sequential(0, 100, int(st[0]))   # warm, one
        _, dt_one = between_barriers(
            lambda: chains[0].sweep_sequential(100, 100 + rows_c, st1))
        st = _core.sweep_sequential_many(cores_, 100 + rows_c,
                                         200 + rows_c, st)    # warm, all
        first = 200 + rows_c
        rows_m = min(rows_c, (n_c - first) // 2)
        # two chains per compute unit (FastLog's table in LDS) ...
        m_half = min(m, 512)
        head, dt_half = between_barriers(
            lambda: _core.sweep_sequential_many(
                cores_[:m_half], first, first + rows_m, st[:m_half]))
        st[:m_half] = head
        # ... and all of them (beyond 512: four per CU, the table in L2)
        _, dt_many = between_barriers(
            lambda: _core.sweep_sequential_many(
                cores_, first + rows_m, first + 2 * rows_m, st))
        exact = {"sequential_value": world * rows_c / dt_one,
                 "sequential_us_per_row": 1e6 * dt_one / rows_c,
                 "sequential_cycles_per_row": CLOCK_GHZ * 1e9 * dt_one / rows_c,
                 "exact_chains_value": world * m * rows_m / dt_many,
                 "chains": world * m, "chains_per_gpu": m,
                 "rows_per_chain": rows_m,
                 "us_per_row_and_chain": 1e6 * dt_many / rows_m,
                 "exact_chains_512_value": world * m_half * rows_m / dt_half,
                 "unit": "row-updates/s",
                 "note": "the reference's sequential sampler itself, "
                         "bit-exact per chain against the oracle "
                         "(tests/test_gpu_chains.py): k_chains, one "
                         "workgroup per chain, group creation and removal "
                         "on the device; %s K=%d+1, %d rows per chain"
                         % (name, k, n_c)}
        del chains, cores_
        torch.cuda.empty_cache()

    strong = None
    if world > 1 and not args.no_strong:
        # strong scaling: the SAME N rows in total, split over the ranks
        comm = sharded.native_comm
        del sharded, g, columns
        torch.cuda.empty_cache()
        n_s = n // world
        g2, sharded2, _, _, _ = build_job(n_s, rank * n_s)
        if native_comm:
            sharded2.use_native_comm(comm)
        if by_value:
            sharded2.partition_by_value()
        steps_s = max(1, min(args.steps, 10))
        batch_s = max(1, min(args.batch, n_s))
        dt_s = timed(sharded2, g2, n_s, batch_s, steps_s, args.warmup, 0)
        strong = {"value": float(n_s) * world * steps_s / dt_s,
                  "unit": "row-updates/s", "rows_total": n_s * world,
                  "rows_per_gpu": n_s, "batch_rows": batch_s,
                  "steps": steps_s, "ms_per_step": 1e3 * dt_s / steps_s}

    total_rows = float(n) * world * args.steps
    if vs_batches and scanned:
        kernel = "k_vs_scan_rows<%s>" % args.config
    elif scratched and "sampling=1" in [o.replace(" ", "") for o in args.opt]:
        kernel = "k_rows_scratch<%s, scan>" % args.config
    elif vs_batches and streamed:
        kernel = "k_vs_stream<%s>" % args.config
    elif vs_batches and narrow:
        kernel = "k_vs_narrow<%s>" % args.config
    elif vs_batches:
        kernel = "k_vs_sample<%s>" % args.config
    elif scratched:
        kernel = "k_rows_scratch<%s>" % args.config
    elif args.config == "mixed":
        kernel = "k_sweep_program"
    else:
        kernel = "k_sweep_sample<%s>" % args.config
    if rank == 0:
        launches = max(launches, 1)
        avg_ms = ms / launches
        rows_per_launch = rows / launches
        ctr = committed_counters(kernel)
        usable = ctr is not None and not ctr["stale"]
        scale = (rows_per_launch / ctr["rows_per_launch"]) if usable else 0.0
        traffic = ((ctr["fetch_bytes"] + ctr["write_bytes"]) * scale
                   if usable and ctr.get("fetch_bytes") is not None else None)
        valu_cycles = (ctr["valu_busy_cycles"] * scale
                       if usable and ctr.get("valu_busy_cycles") is not None
                       else None)
        secs = 1e-3 * avg_ms
        if secs <= 0.0:   # (--kernel-timing 0: nothing was measured)
            secs = float("nan")
            traffic = valu_cycles = None
        hbm_frac = (traffic / secs / 1e9 / HBM_PEAK_GBS
                    if traffic is not None else None)
        valu_frac = (valu_cycles / (SIMDS * CLOCK_GHZ * 1e9 * secs)
                     if valu_cycles is not None else None)
        # the roof the kernel sits closer to, by the counters (every kernel
        # measured so far is VALU-bound: C5, the configuration SURVEY 8d
        # expected to be HBM-bound, included -- DESIGN.md section 4)
        hbm_bound = (hbm_frac is not None and valu_frac is not None
                     and hbm_frac > valu_frac)
        if hbm_bound:
            roof = {"bound": "hbm", "kernel": kernel,
                    "achieved": traffic / secs / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
                    "valu_frac": valu_frac}
        else:
            roof = {"bound": "valu", "kernel": kernel,
                    "achieved": (valu_cycles / secs / 1e9
                                 if valu_cycles is not None else None),
                    "peak": SIMDS * CLOCK_GHZ,
                    "unit": "G SIMD-busy-cycles/s", "frac": valu_frac,
                    "hbm_frac": hbm_frac}
        # essential / issued vector instructions of the value-sorted kernel:
        # what the exact algorithm cannot do without is one packed add per
        # entry, tile of 128 rows and pass, the total's pass starting at the
        # tile's first own chunk (3/4 of K on average): 2 x K x 0.75
        useful = None
        if (usable and ctr.get("valu_instructions") and vs_batches
                and not streamed and not narrow):
            tiles = rows_per_launch / 128.0
            useful = (1.5 * k * tiles) / (ctr["valu_instructions"] * scale)
        roof["useful_frac"] = useful
        # the same busy fraction with the instructions priced by class
        # instead of 4 cycles each: the packed adds the algorithm needs
        # (counted analytically, as for useful_frac) at the packed cost, every
        # other vector instruction at the plain-operation cost -- compares and
        # shifts among them cost more, so this is a lower bound on the issue
        # utilisation as `frac` is an upper one
        weighted = None
        if useful is not None and secs == secs:
            packed = 1.5 * k * (rows_per_launch / 128.0)
            other = max(ctr["valu_instructions"] * scale - packed, 0.0)
            weighted = ((packed * ISSUE_CYCLES_PACKED
                         + other * ISSUE_CYCLES_PLAIN)
                        / (SIMDS * CLOCK_GHZ * 1e9 * secs))
        roof["frac_weighted"] = weighted
        roof.update({
            "traffic": traffic,
            "algorithmic_bytes_per_row": bytes_per_row,
            "algorithmic_GBps": (bytes_per_row * rows_per_launch / secs / 1e9
                                 if secs == secs else None),
            "rows_per_launch": rows_per_launch,
            "avg_launch_ms": avg_ms if secs == secs else None,
            "launches": launches,
            "counters": (None if ctr is None else
                         {"file": "profiles/" + COUNTERS_FILE,
                          "stale": ctr["stale"],
                          "rows_per_launch": ctr["rows_per_launch"]}),
            # non-null: why `frac` is missing -- never silently
            "stale_reason": (
                None if usable else
                "no counters for %s in profiles/%s" % (kernel, COUNTERS_FILE)
                if ctr is None else
                "the kernel sources changed since profiles/%s was taken "
                "(hash %s): run tools/profile_round.sh on a GPU box"
                % (COUNTERS_FILE, source_hash())),
            "timed_every": args.kernel_timing,
            "note": "avg_launch_ms: HIP events on the launch stream around "
                    "every `timed_every`-th launch of the timed region, this "
                    "run.  frac: VALU-busy cycles (SQ_ACTIVE_INST_VALU x 4) "
                    "over 1024 SIMDs x 2.4 GHz x kernel time for the "
                    "VALU-bound kernels, HBM bytes (FETCH_SIZE + WRITE_SIZE) "
                    "over time over 8 TB/s for the HBM-bound one; counters "
                    "from separate rocprofv3 --pmc passes of this command "
                    "(profiles/).  algorithmic_GBps is SURVEY 8d's "
                    "streaming-formulation byte count over the kernel time: "
                    "the value-sorted kernels do not move those bytes (one "
                    "likelihood vector serves 128 rows), so it is not a "
                    "fraction of any roof: DESIGN.md section 4",
        })
        if roof.get("stale_reason"):
            sys.stderr.write("bench.py: WARNING roofline.frac is missing: %s\n"
                             % roof["stale_reason"])
        out = {
            "metric": METRIC,
            "value": total_rows / dt,
            "unit": "row-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            # (the host's share: what queueing a step's launches took it)
            "host_enqueue_ms_per_step": host_enqueue_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s N=%d rows/GPU K=%d+1 %s values "
                            "PitmanYor(alpha=%g,d=%g), frozen sub-sweeps of "
                            "%d rows" % (name, n, k, args.values, args.alpha,
                                         args.d, args.batch),
                "rows_per_gpu": n, "groups": k, "dim": args.dim,
                "batch_rows": args.batch,
                "groups_at_end_of_timed_region": groups_at_end,
                "parallelism": "rows sharded over %d GPU(s), all-reduce of "
                               "statistic deltas per sub-sweep" % world,
                "collectives": ("none" if not sharded_collective(
                                    world, args) else
                                "library RCCL communicator" if native_comm
                                else "torch.distributed (RCCL)"),
                "comm_ranks": world if sharded_collective(world, args) else 0,
                "group_set": ("normalised on the device, the host's copy "
                              "pulled once at the end of the timed region"
                              if timed_counts["device_normalised"]
                              else "normalised by the host after every "
                                   "sub-sweep"),
                "launches_per_sub_sweep": (
                    "4: k_vs_tables, score+sample, k_vs_apply, k_vs_reduce"
                    if timed_counts["fused_batches"] else "separate launches"),
                # the other BASELINE configurations of this run as scalars
                # (row-updates/s, exact mode; the records are in
                # `other_configs` / `batch_variants`)
                "b65536_value": next((v["value"] for v in variants
                                      if v["batch_rows"] == 65536), None),
                "b65536_fresh_chain_value": next(
                    (v.get("fresh_chain_value") for v in variants
                     if v["batch_rows"] == 65536), None),
                "c3_value": next((o["value"] for o in others
                                  if o["config"] == "gp_nich"
                                  and o["sampling"] == "exact"), None),
                "c5_value": next((o["value"] for o in others
                                  if o["config"] == "dpd"
                                  and o["sampling"] == "exact"), None),
                "mixed_value": next((o["value"] for o in others
                                     if o["config"] == "mixed"
                                     and o["sampling"] == "exact"), None),
            },
            # the job running on for seconds (see --sustained-seconds)
            "sustained_value": sustained["value"] if sustained else None,
            "sustained": sustained,
            # the reference's own sampler on the device (see --exact-chains)
            "sequential_value": exact["sequential_value"] if exact else None,
            "exact_chains_value": (exact["exact_chains_value"]
                                   if exact else None),
            "exact_chains": exact,
            "roofline": roof,
            "step_breakdown": breakdown,
            "batch_variants": variants,
            "other_configs": others,
        }
        if sharded_collective(world, args):
            import torch.distributed as dist
            # (the communicator the timed region ran on is the job's: a run
            # that fell back to fewer ranks is not an N-GPU number)
            assert dist.get_world_size() == world == max(args.gpus, 1) or \
                args.force_collective, "communicator size != --gpus"
            sub_sweeps = -(-n // args.batch)
            avg_us = 1e3 * comm_ms / comm_count if comm_count else None
            step_us = 1e6 * dt / args.steps
            out["comm"] = {
                "ranks": world,
                "all_reduce_avg_us": avg_us,
                # nothing runs under the collective (every kernel
                # of the next sub-sweep needs its result), so all of it is
                # exposed: its share of the step, and what the step would
                # take without it
                "sub_sweeps_per_step": sub_sweeps,
                "all_reduce_us_per_step": (avg_us * sub_sweeps
                                           if avg_us is not None else None),
                "exposed_fraction_of_step": (avg_us * sub_sweeps / step_us
                                             if avg_us is not None else None),
                "weak_scaling_budget_us": {
                    # 8 * t / (t + L) >= 6 with t = the sub-sweep without
                    # the collective
                    "sub_sweep_without_all_reduce_us": (
                        (step_us - avg_us * sub_sweeps) / sub_sweeps
                        if avg_us is not None else None),
                    "max_all_reduce_us_for_6x_at_8": (
                        (step_us - avg_us * sub_sweeps) / sub_sweeps / 3.0
                        if avg_us is not None else None)},
                "timed": comm_count,
                # counted by dist_gibbs_sweep_sharded over the timed region
                # itself (4 header words included): the live part of the
                # group set, or 3 words per group on value-partitioned ranks
                "placement": "value" if by_value else "block",
                "all_reduces_in_run": volume["collectives"] if volume else None,
                "words_in_run": volume["words"] if volume else None,
                "words_per_all_reduce": (
                    volume["words"] / max(volume["collectives"], 1)
                    if volume else None),
                "words_largest_all_reduce": (volume["words_max"]
                                             if volume else None),
                "bytes_per_all_reduce": (
                    4.0 * volume["words"] / max(volume["collectives"], 1)
                    if volume else None),
                "note": "HIP events around the library's in-place all-reduce "
                        "of the int32 delta image, every `timed_every`-th "
                        "sub-sweep of the timed region and the variants; "
                        "words_*: what those all-reduces carried, counted "
                        "by the library"}
        if strong is not None:
            out["strong_scaling"] = strong
        if column_host is not None:
            out["cpu_baseline"] = cpu_baseline(args, column_host)
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
    return 0


def sharded_collective(world, args):
    return world > 1 or args.force_collective


def main():
    args = parse()
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
