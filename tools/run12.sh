timeout 1500 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_edges.py tests/test_gpu_two_ranks.py -x -q 2>&1 | tail -3
run() { cfg=$1; shift; args=""; for kv in "$@"; do args="$args --opt $kv"; done
  echo "== $cfg $*"
  timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-rows 0 --other-batches "" --other-configs "" --kernel-timing 1 $args 2>&1 | tail -1 | python tools/brief.py; }
run mixed apply_stage=0
run mixed
run mixed sampling=1
run dd16 value_sorted=0
run dd16 value_sorted=0 apply_stage=0
