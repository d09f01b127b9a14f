#!/bin/bash
# kernel durations of the bench at three shapes, quickly (rocprofv3 --stats):
# tools/quick_trace.sh -> gpurun_out/quick_stats_{c2,c2batch65536,c2valueszipf}.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --cpu-rows 0 --other-batches= --other-configs= --no-breakdown --exact-chains 0 --sustained-seconds 0"
for v in "" "--batch 65536" "--values zipf"; do
  name=c2$(echo $v | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_$name -- python3 $B --steps 10 --warmup 3 $v > gpurun_out/quick_bench_$name.json 2>/dev/null
  f=$(ls gpurun_out/q_$name/*/*kernel_stats.csv | head -1)
  cp $f gpurun_out/quick_stats_$name.csv
  rm -rf gpurun_out/q_$name
  python3 tools/quick_trace_summary.py $name
done
