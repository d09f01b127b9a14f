#!/bin/bash
# The round's evidence, on the GPU box: tools/profile_round.sh <tag>
# writes gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   kernel trace + stats of the bench command, the SQ / FETCH_SIZE / WRITE_SIZE
#   counter passes (each its own run, --kernel-trace only beside --pmc),
#   counters.json (tools/counters.py), the bench line with the CPU baseline
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp && cd $repo
B="bench.py --cpu-rows 0 --other-batches="
SQ="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $B > $out/bench_under_rocprof.json 2> $out/trace.log
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
# the same trace restricted to the timed region (the last 20 sweeps' launches:
# kernel_stats.csv averages the two warm-up sweeps in, whose first runs on
# tiles not yet sorted by group)
python3 - $out <<'PY' > $out/kernel_stats_timed_region.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"].split("(")[0], []).append(
        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel, launches in the timed region (last 200 of each per-sub-sweep kernel), average us")
for name, d in sorted(by.items(), key=lambda kv: -sum(kv[1][-200:])):
    if len(d) < 200:
        continue
    last = d[-200:]
    print("%-60s %4d %8.2f" % (name[:60], len(last), sum(last) / len(last) / 1e3))
PY
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sq -- python3 $B --steps 3 --warmup 2 > /dev/null 2> $out/sq.log
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/grbm -- python3 $B --steps 3 --warmup 2 > /dev/null 2> $out/grbm.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 $B --steps 3 --warmup 2 > /dev/null 2> $out/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 $B --steps 3 --warmup 2 > /dev/null 2> $out/write.log
# C5 (the HBM-bound configuration) and C3 (the general-row kernel)
C5="--config dpd --groups 8192 --dim 10000 --steps 3 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -- python3 $B $C5 > $out/bench_c5_under_rocprof.json 2>/dev/null
cp $(ls $out/c5/*/*kernel_stats.csv | head -1) $out/kernel_stats_c5_dpd.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/c5f -- python3 $B $C5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/c5w -- python3 $B $C5 > /dev/null 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/c5s -- python3 $B $C5 > /dev/null 2>&1
C3="--config gp_nich --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3 -- python3 $B $C3 > $out/bench_c3_under_rocprof.json 2>/dev/null
cp $(ls $out/c3/*/*kernel_stats.csv | head -1) $out/kernel_stats_c3_gp_nich.csv
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/c3s -- python3 $B $C3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/c3f -- python3 $B $C3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/c3w -- python3 $B $C3 > /dev/null 2>&1
# the same workload in sub-sweeps of 65 536 rows (k_vs_narrow)
SM="--batch 65536 --steps 2 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/sm -- python3 $B $SM > $out/bench_b65536_under_rocprof.json 2>/dev/null
cp $(ls $out/sm/*/*kernel_stats.csv | head -1) $out/kernel_stats_b65536.csv
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $out/sms -- python3 $B $SM > /dev/null 2>&1
python3 tools/counters.py $out/counters.json \
    "k_vs_sample<dd>=k_vs_sample<0, 1024>:1000000:100000" \
    "k_vs_narrow<dd>=k_vs_narrow<0, 8>:65536:60000" \
    "k_vs_stream<dpd>=k_vs_stream<4>:1000000:100000" \
    "k_sweep_sample<gp_nich>=k_sweep_sample<2, 3, 2>:1000000:100000" \
    -- $out/sq $out/grbm $out/fetch $out/write $out/sms $out/c5f $out/c5w $out/c5s $out/c3s $out/c3f $out/c3w > $out/counters.log 2>&1
for d in sq grbm fetch write sms c5f c5w c5s c3s c3f c3w; do
  python3 tools/pmc_summary.py $out/$d k_ > $out/pmc_$d.txt 2>/dev/null
done
# the bench line of record (with the CPU baseline), the other configurations,
# the VALU issue-rate microbenchmark
python3 bench.py 2> $out/bench.log | tail -1 > $out/bench.json
: > $out/bench_other_configs.jsonl
for c in gp_nich nich gp bb mixed dd16; do
  python3 bench.py --cpu-rows 0 --other-batches= --steps 3 --warmup 1 --config $c 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
done
python3 bench.py --cpu-rows 0 --other-batches= --steps 3 --warmup 2 --config dpd --groups 8192 --dim 10000 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
python3 bench.py --cpu-rows 0 --other-batches= --steps 5 --values zipf 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
python3 bench.py --cpu-rows 0 --other-batches= --steps 5 --d 0 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
python3 bench.py --cpu-rows 0 --other-batches= --force-collective 2>/dev/null | tail -1 > $out/bench_collective_one_rank.json
python3 bench.py --cpu-rows 0 --other-batches= --device-normalise 1 2>/dev/null | tail -1 > $out/bench_device_normalise.json
(cd tools/microbench && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o valu_issue valu_issue.hip && ./valu_issue) > $out/valu_issue.txt 2>&1
rm -rf $out/sm $out/sms $out/trace $out/sq $out/grbm $out/fetch $out/write $out/c5 $out/c5f $out/c5w $out/c5s $out/c3 $out/c3s $out/c3f $out/c3w
ls -la $out
