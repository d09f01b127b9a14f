# what the per-batch timing events cost: tools/ab_timing.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for M in 1 8 0; do
  echo -n "kernel_timing=$M " >> gpurun_out/kt_ab.txt
  timeout 300 python bench.py --no-strong --other-batches "" --cpu-rows 0 --kernel-timing $M 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['roofline']['frac'])" >> gpurun_out/kt_ab.txt
done
done
