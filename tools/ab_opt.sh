# A/B of one engine option on the headline line, alternating, twice each:
#   bash tools/ab_opt.sh tile_balance 1 0 [extra bench.py arguments]
opt=$1; a=$2; b=$3; shift 3
B="python bench.py --steps 20 --warmup 5 --no-breakdown --other-batches= --other-configs= --cpu-rows 8192 $*"
for rep in 1 2; do
for o in $a $b; do
  $B --opt $opt=$o 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$opt=$o', '%.3f G/s' % (d['value']/1e9), '%.4f ms/step' % d['ms_per_step'], 'kernel %.2f us' % (1e3*d['roofline'].get('avg_launch_ms', 0)))
"
done
done
