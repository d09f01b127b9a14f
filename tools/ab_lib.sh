# A/B of two BUILDS of the library on one box, alternating: copy the build to compare against to
# distributions_amd/libdist_base.so (git-ignored) before the gpurun call, then
#   [CFGS="dd gp"] [STEPS=20] [WARMUP=5] bash tools/ab_lib.sh [bench.py arguments]
L=distributions_amd/libdistributions_hip.so
cp $L /tmp/new.so; cp distributions_amd/libdist_base.so /tmp/base.so
run() { python bench.py --steps ${STEPS:-4} --warmup ${WARMUP:-2} --no-breakdown --other-batches= --other-configs= --cpu-rows 8192 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%.4f G/s  %.3f ms/step' % (d['value']/1e9, d['ms_per_step']))
"; }
for cfg in ${CFGS:-gp_nich gp nich mixed}; do
  for rep in 1 2; do
    cp /tmp/base.so $L; echo -n "$cfg base: "; run --config $cfg "$@"
    cp /tmp/new.so $L; echo -n "$cfg new:  "; run --config $cfg "$@"
  done
done
cp /tmp/new.so $L
