timeout 1200 python -m pytest tests/test_gpu_scan.py -x -q -s 2>&1 | grep -v Warning | tail -40
run() { cfg=$1; shift; args=""; for kv in "$@"; do args="$args --opt $kv"; done
  echo "== $cfg $*"
  timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-rows 0 --other-batches "" --kernel-timing 1 $args 2>&1 | tail -1 | python tools/brief.py; }
run gp_nich sampling=1
run gp_nich sampling=1 rows_scratch_block=768
run gp_nich sampling=1 rows_scratch_block=1024
run gp_nich sampling=1 rows_fold=0
run mixed sampling=1
run nich sampling=1
run gp value_sorted=0 sampling=1
run dd16 value_sorted=0 sampling=1
