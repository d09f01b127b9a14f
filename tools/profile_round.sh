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
python3 tools/counters.py $out/counters.json \
    "k_vs_sample<dd>=k_vs_sample<0>:1000000:100000" \
    "k_vs_sample<dpd>=k_vs_sample<4>:1000000:100000" \
    "k_vs_prepare<dpd>=k_vs_prepare<4>:1000000:0" \
    "k_sweep_sample<gp_nich>=k_sweep_sample<2, 3, 2>:1000000:100000" \
    -- $out/sq $out/grbm $out/fetch $out/write $out/c5f $out/c5w $out/c5s $out/c3s $out/c3f $out/c3w > $out/counters.log 2>&1
for d in sq grbm fetch write c5f c5w c5s c3s c3f c3w; do
  python3 tools/pmc_summary.py $out/$d k_ > $out/pmc_$d.txt 2>/dev/null
done
rm -rf $out/trace $out/sq $out/grbm $out/fetch $out/write $out/c5 $out/c5f $out/c5w $out/c5s $out/c3 $out/c3s $out/c3f $out/c3w
ls -la $out
