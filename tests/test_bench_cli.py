"""bench.py's launcher contract: `--gpus N` starts N ranks itself, refuses to
report an N-GPU number from fewer devices, and rejects a world size that
disagrees with --gpus."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, text=True,
                          capture_output=True, timeout=timeout)


def gpu_count():
    import torch
    return torch.cuda.device_count()


def test_more_gpus_than_devices_is_refused():
    ask = max(2, gpu_count() + 1)
    r = run(["--gpus", str(ask), "--cpu-rows", "0"])
    assert r.returncode != 0
    assert "refusing" in r.stderr and "--gpus %d" % ask in r.stderr
    assert r.stdout.strip() == ""          # no JSON line from a refused run


def test_world_size_must_agree_with_gpus():
    r = run(["--gpus", "2"], env={"WORLD_SIZE": "4", "RANK": "0",
                                  "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


@pytest.mark.gpu
def test_small_single_gpu_run_prints_one_json_line():
    r = run(["--rows", "400000", "--batch", "100000", "--steps", "2",
             "--warmup", "1", "--cpu-rows", "20000", "--other-batches",
             "50000", "--exact-chains", "8", "--exact-rows", "500",
             "--sustained-seconds", "0.2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    # the job running on, and the reference's own sampler on the device
    assert out["sustained_value"] > 0
    assert out["sustained"]["seconds"] >= 0.2
    assert out["sustained"]["groups_at_end"] > 0
    assert out["sequential_value"] > 0 and out["exact_chains_value"] > 0
    assert out["exact_chains"]["chains"] == 8
    assert out["batch_variants"][0]["fresh_chain_value"] > 0
    # a missing roofline fraction is never silent
    assert (out["roofline"]["frac"] is not None
            or out["roofline"]["stale_reason"])
    assert out["n_gpus"] == 1 and out["unit"] == "row-updates/s"
    assert out["value"] > 0 and out["cpu_baseline"]["value"] > 0
    assert out["cpu_baseline"]["cores"] == 1
    assert out["batch_variants"][0]["batch_rows"] == 50000
    roof = out["roofline"]
    assert roof["bound"] in ("valu", "hbm") and roof["avg_launch_ms"] > 0
    for key in ("frac", "frac_weighted", "hbm_frac", "valu_frac"):
        if roof.get(key) is not None:
            assert 0.0 < roof[key] <= 1.0, (key, roof[key])
    assert roof["kernel"].startswith(("k_vs_sample", "k_vs_narrow"))
    # what the line says about the run is the timed engine's own account
    assert "normalised on the device" in out["config"]["group_set"]
    assert out["config"]["launches_per_sub_sweep"].startswith("4:")
    # the other BASELINE configurations, exact and with scan sampling
    got = [(o["config"], o["sampling"].split()[0]) for o in out["other_configs"]]
    assert got == [("gp_nich", "exact"), ("gp_nich", "scan"),
                   ("gp_nich", "scan"), ("mixed", "exact"), ("mixed", "scan"),
                   ("mixed", "scan"), ("dpd", "exact"), ("dpd", "scan")]
    for o in out["other_configs"]:
        assert o["value"] > 0
        scan = o["sampling"].startswith("scan")
        assert ("scan" in o["kernel"]) == scan, o   # a scan run names its kernel
        if o["config"] != "dpd":
            assert o["kernel"].startswith("k_rows_scratch")
        else:
            assert o["groups"] == 8192
    cfg = out["config"]
    assert cfg["c3_value"] == out["other_configs"][0]["value"]
    assert cfg["c5_value"] == out["other_configs"][6]["value"]
    assert cfg["b65536_value"] is None     # (this run's variant is 50 000)
    assert out["cpu_baseline"]["reference_kernels"] is None or (
        out["cpu_baseline"]["reference_kernels"]["K=1024"]
        ["vector_exp_elements_per_us"] > 0)


@pytest.mark.gpu
def test_one_rank_collective_path_reports_its_all_reduce():
    r = run(["--rows", "400000", "--batch", "100000", "--steps", "3",
             "--warmup", "1", "--cpu-rows", "0", "--other-batches", "",
             "--force-collective", "--kernel-timing", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([x for x in r.stdout.splitlines()
                      if x.startswith("{")][-1])
    assert out["config"]["comm_ranks"] == 1
    assert out["config"]["collectives"] == "library RCCL communicator"
    assert out["comm"]["timed"] >= 1 and out["comm"]["all_reduce_avg_us"] > 0
    assert "normalised on the device" in out["config"]["group_set"]
    # what the exchanges carried, counted by the library: value-partitioned
    # ranks send 3 words per live group (+ the 4-word header)
    comm = out["comm"]
    assert comm["placement"] == "value"
    assert comm["all_reduces_in_run"] == 3 * 4
    assert 4 + 3 * 1025 <= comm["words_per_all_reduce"] <= 4 + 3 * 1200
    assert comm["words_largest_all_reduce"] <= 4 + 3 * 1200


@pytest.mark.gpu
def test_one_rank_collective_path_block_placement():
    r = run(["--rows", "400000", "--batch", "100000", "--steps", "2",
             "--warmup", "1", "--cpu-rows", "0", "--other-batches", "",
             "--force-collective", "--placement", "block"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([x for x in r.stdout.splitlines()
                      if x.startswith("{")][-1])
    comm = out["comm"]
    assert comm["placement"] == "block"
    # the LIVE part of the group set, never the run's bound (8128 groups)
    assert (4 + 259 * 1025 <= comm["words_per_all_reduce"]
            <= 4 + 259 * 1200)


@pytest.mark.gpu
@pytest.mark.skipif(gpu_count() < 2, reason="needs two GPUs")
def test_two_ranks_over_rccl():
    """--gpus 2: two processes, one RCCL communicator of two ranks, the
    library's own all-reduce per sub-sweep; plus the strong-scaling leg"""
    r = run(["--gpus", "2", "--rows", "400000", "--batch", "100000",
             "--steps", "2", "--warmup", "1", "--cpu-rows", "0",
             "--other-batches", ""])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([x for x in r.stdout.splitlines()
                      if x.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["comm_ranks"] == 2
    assert out["config"]["collectives"] == "library RCCL communicator"
    assert out["strong_scaling"]["rows_total"] == 400000
