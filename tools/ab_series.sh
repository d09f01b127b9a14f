#!/bin/bash
# A kernel's durations over the bench run under two settings of one option:
#   bash tools/ab_series.sh debug.apply_overlap k_vs_apply [bench.py arguments]
opt=$1; kern=$2; shift 2
repo=$PWD
cd /tmp && export TMPDIR=/tmp && cd $repo
for o in 1 0; do
  rm -rf gpurun_out/abs_$o
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abs_$o -- python3 bench.py --steps 20 --warmup 5 --no-breakdown --other-batches= --other-configs= --cpu-rows 0 --opt $opt=$o "$@" > /dev/null 2>&1
  echo "$opt=$o"
  python3 tools/kernel_series.py gpurun_out/abs_$o $kern 10 60
  rm -rf gpurun_out/abs_$o
done
