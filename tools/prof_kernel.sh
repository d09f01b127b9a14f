#!/bin/bash
# Counter passes of ONE bench command for ONE kernel, on the GPU box:
#   tools/prof_kernel.sh <tag> <kernel-substring> <bench.py arguments...>
# writes gpurun_out/<tag>/{kernel_stats.csv,pmc.txt}: the kernel trace's stats
# and the per-launch means of the SQ / FETCH_SIZE / WRITE_SIZE passes (each
# its own run: --kernel-trace only beside --pmc).
set -u
tag=$1; shift
sub=$1; shift
out=gpurun_out/$tag
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp && cd $repo
B="bench.py --cpu-rows 0 --other-batches= --steps 2 --warmup 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $B > $out/bench_under_rocprof.json 2> $out/trace.log
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
SQ1="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAVES"
: > $out/pmc.txt
for pass in "$SQ1" "$SQ2" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  rm -rf $out/p
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/p -- python3 $B > /dev/null 2> $out/p.log
  python3 tools/pmc_summary.py $out/p "$sub" >> $out/pmc.txt 2>/dev/null
done
rm -rf $out/p $out/trace
python3 tools/kstats.py $out 2>/dev/null || head -8 $out/kernel_stats.csv
cat $out/pmc.txt
