cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --cpu-rows 0 --other-batches= --other-configs= --no-breakdown --steps 3 --warmup 2 --config dpd --groups 8192 --dim 10000"
for i in 1 2; do python3 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['ms_per_step'], d['roofline'].get('avg_launch_ms'))"; done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t_c5 -- python3 $B > /dev/null 2>&1
python3 tools/batch_timeline.py gpurun_out/t_c5 k_vs_stream
rm -rf gpurun_out/t_c5
timeout 600 python3 tools/fuzz.py 300 50000 2>&1 | tail -1
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
