cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --cpu-rows 0 --other-batches= --other-configs= --no-breakdown --exact-chains 0 --sustained-seconds 0"
for v in "" "--batch 65536"; do
  name=c2$(echo $v | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_$name -- python3 $B --steps 10 --warmup 3 $v > /dev/null 2>&1
  f=$(ls gpurun_out/q_$name/*/*kernel_stats.csv | head -1)
  echo "== $name"; head -8 $f | cut -d, -f1-8
  cp $f gpurun_out/r6_quick_stats_$name.csv
  rm -rf gpurun_out/q_$name
done
