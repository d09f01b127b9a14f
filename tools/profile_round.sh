#!/bin/bash
# The round's evidence, on the GPU box: tools/profile_round.sh <tag>
# writes gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   kernel trace + stats of the bench command, the SQ / FETCH_SIZE / WRITE_SIZE
#   counter passes (each its own run, --kernel-trace only beside --pmc),
#   counters.json (tools/counters.py), one sub-sweep's launches in order
#   (tools/batch_timeline.py), the bench line with the CPU baseline
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp && cd $repo
B="bench.py --cpu-rows 0 --other-batches= --other-configs= --no-breakdown --exact-chains 0 --sustained-seconds 0"
SQ="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"

# one configuration: trace (+ timeline of a sub-sweep), SQ / FETCH / WRITE
# passes; $1 = name, $2 = kernel that starts a sub-sweep, rest = bench args
# STEPS / WARMUP: the c2 line is profiled over the very run the driver times
# (20 + 5 sweeps: the group count creeps up through it, and with it the
# kernels' durations); the other configurations over 3 + 2
profile() {
  name=$1; anchor=$2; shift 2
  steps=${STEPS:-3}; warm=${WARMUP:-2}
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$name -- python3 $B --steps $steps --warmup $warm "$@" > $out/bench_${name}_under_rocprof.json 2> $out/t_$name.log
  cp $(ls $out/t_$name/*/*kernel_stats.csv | head -1) $out/kernel_stats_$name.csv
  rm -rf $out/t_$name
  # (the sequence of one sub-sweep from a trace of its own: --stats perturbs)
  rocprofv3 --kernel-trace --output-format csv -d $out/t_$name -- python3 $B --steps $steps --warmup $warm "$@" > /dev/null 2>&1
  python3 tools/batch_timeline.py $out/t_$name "$anchor" > $out/timeline_$name.txt 2>/dev/null
  # ... and every launch of the sub-sweep's kernels over the run (what drifts)
  : > $out/series_$name.txt
  for k in k_vs_tables k_vs_sample k_vs_narrow k_vs_stream k_vs_apply k_vs_reduce k_rows_scratch k_replay_sorted k_cs_; do
    python3 tools/kernel_series.py $out/t_$name $k >> $out/series_$name.txt 2>/dev/null
  done
  for pass in SQ FETCH_SIZE WRITE_SIZE; do
    ctr=$pass; [ $pass = SQ ] && ctr="$SQ"
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/p_${name}_$pass -- python3 $B --steps $steps --warmup $warm "$@" > /dev/null 2> $out/p.log
    python3 tools/pmc_summary.py $out/p_${name}_$pass k_ > $out/pmc_${name}_$pass.txt 2>/dev/null
  done
  rm -rf $out/t_$name $out/t_$name.log
}
STEPS=20 WARMUP=5 profile c2 k_vs_tables
profile c2_b65536 k_vs_tables --batch 65536
profile c2_zipf k_vs_tables --values zipf
profile c2_unfused k_vs_prepare --opt fused_tables=0
profile c2_scan k_vs_scan_prepare --opt sampling=1
profile c3 k_rows_scratch --config gp_nich
profile c3_scan k_rows_scratch --config gp_nich --opt sampling=1 --opt float_stats=1
profile mixed k_rows_scratch --config mixed
profile c5 k_vs_stream --config dpd --groups 8192 --dim 10000
profile c5_scan k_vs_scan_prepare --config dpd --groups 8192 --dim 10000 --opt sampling=1
python3 tools/counters.py $out/counters.json \
    "k_vs_sample<dd>=k_vs_sample<0, 512>:1000000:100000" \
    "k_vs_narrow<dd>=k_vs_narrow<0, 8>:65536:60000" \
    "k_vs_stream<dpd>=k_vs_stream<4>:1000000:100000" \
    "k_rows_scratch<gp_nich>=k_rows_scratch<false, true, 3>:1000000:100000" \
    "k_rows_scratch<gp_nich, scan>=k_rows_scratch<true, true, 3>:1000000:100000" \
    "k_vs_scan_rows<dd>=k_vs_scan_rows<0>:1000000:100000" \
    "k_vs_scan_rows<dpd>=k_vs_scan_rows<4>:1000000:100000" \
    -- $out/p_c2_SQ $out/p_c2_FETCH_SIZE $out/p_c2_WRITE_SIZE \
       $out/p_c2_b65536_SQ $out/p_c5_SQ $out/p_c5_FETCH_SIZE $out/p_c5_WRITE_SIZE \
       $out/p_c3_SQ $out/p_c3_FETCH_SIZE $out/p_c3_WRITE_SIZE \
       $out/p_c3_scan_SQ $out/p_c3_scan_FETCH_SIZE $out/p_c3_scan_WRITE_SIZE \
       $out/p_c2_scan_SQ $out/p_c2_scan_FETCH_SIZE $out/p_c2_scan_WRITE_SIZE \
       $out/p_c5_scan_SQ $out/p_c5_scan_FETCH_SIZE $out/p_c5_scan_WRITE_SIZE > $out/counters.log 2>&1
rm -rf $out/p_* $out/p.log
# bench.py looks the counters up under profiles/ (keyed by the sources' hash)
cp $out/counters.json profiles/${tag}_counters.json
# the bench line of record (the driver's command, with the CPU baseline), the
# other configurations
python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $out/bench.log | tail -1 > $out/bench.json
: > $out/bench_other_configs.jsonl
for c in nich gp bb dd16; do
  python3 $B --steps 3 --warmup 1 --config $c 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
done
python3 $B --steps 3 --warmup 2 --config dpd --groups 8192 --dim 10000 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
python3 $B --steps 3 --warmup 2 --config dpd --groups 8192 --dim 10000 --opt sampling=1 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
python3 $B --steps 10 --values zipf 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
python3 $B --steps 10 --d 0 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
python3 $B --steps 10 --opt sampling=1 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
python3 $B --steps 10 --batch 65536 --opt sampling=1 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
# the N > 1 code path on one rank: value-partitioned ranks (3 words per group
# and sub-sweep) and block placement (3 + dim words per live group)
python3 $B --steps 20 --warmup 5 --force-collective 2>/dev/null | tail -1 > $out/bench_collective_one_rank.json
python3 $B --steps 20 --warmup 5 --force-collective --placement block 2>/dev/null | tail -1 > $out/bench_collective_one_rank_block.json
python3 $B --steps 3 --warmup 2 --config dpd --groups 8192 --dim 10000 --force-collective 2>/dev/null | tail -1 > $out/bench_collective_one_rank_c5.json
# the exact chains (the reference's own sampler on the device)
python3 tools/exact_chains.py 20000 1 8 64 256 512 1024 > $out/exact_chains.txt 2>&1
ls -la $out
