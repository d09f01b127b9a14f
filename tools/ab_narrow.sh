# A/B of the small-launch kernel: tools/ab_narrow.sh "<batch sizes>" "<narrow_tiles values>"
cd $GRAFT_REPO_ROOT
for B in $1; do
 for M in $2; do
  echo "B=$B narrow=$M" >> gpurun_out/narrow_ab.txt
  timeout 300 python bench.py --no-strong --batch $B --steps 4 --warmup 1 --other-batches "" --cpu-rows 0 --narrow-tiles $M 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'])" >> gpurun_out/narrow_ab.txt
 done
done
