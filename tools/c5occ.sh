# experiment: occupancy of k_vs_stream (C5)
B="bench.py --cpu-rows 0 --other-batches= --other-configs= --no-breakdown --steps 3 --warmup 2 --config dpd --groups 8192 --dim 10000"
run() { python3 $B "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value']/1e9, d['roofline'].get('avg_launch_ms'))"; }
timeout 900 python3 -m pytest tests/test_gpu_sweep.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python3 tools/fuzz.py 300 30000 2>&1 | tail -1
cd distributions_amd/csrc
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fgpu-flush-denormals-to-zero -Wno-unused-function"
for w in 8 6 5 4; do
  /opt/rocm/bin/hipcc $F -DVS_STREAM_WAVES=$w -c -o /tmp/d.o dist_hip.hip 2>/dev/null && /opt/rocm/bin/hipcc $F -shared -o ../libdistributions_hip.so /tmp/d.o sort.o wire.o
  cd ../..
  echo "== waves_per_eu $w"
  case $w in
    8) for pad in 0 ; do run --opt stream_lds_pad=$pad; run --opt stream_lds_pad=$pad --opt stream_scratch=0; done;;
    6) for pad in 0 14848; do run --opt stream_lds_pad=$pad; done;;
    5) for pad in 0 20480; do run --opt stream_lds_pad=$pad; done; run --opt stream_scratch=0;;
    4) for pad in 0 28672; do run --opt stream_lds_pad=$pad; done;;
  esac
  cd distributions_amd/csrc
done
