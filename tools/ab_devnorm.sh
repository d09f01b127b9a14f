# A/B of the device-side normalisation: tools/ab_devnorm.sh "<batch sizes>" <repeats>
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 $2); do
for B in $1; do
 for M in 0 1; do
  echo -n "B=$B devnorm=$M " >> gpurun_out/dn_ab.txt
  timeout 300 python bench.py --no-strong --batch $B --other-batches "" --cpu-rows 0 --device-normalise $M 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> gpurun_out/dn_ab.txt
 done
done
done
