# several values of one option on the headline line, interleaved, twice:
#   bash tools/ab_multi.sh <option> "<v1> <v2> ..." [extra bench.py arguments]
opt=$1; vals=$2; shift 2
B="python bench.py --steps 20 --warmup 5 --no-breakdown --other-batches= --other-configs= --cpu-rows 8192 $*"
for rep in 1 2; do
for o in $vals; do
  $B --opt $opt=$o 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$opt=$o', '%.3f G/s' % (d['value']/1e9), '%.4f ms/step' % d['ms_per_step'], 'kernel %.2f us' % (1e3*d['roofline'].get('avg_launch_ms', 0)))
"
done
done
