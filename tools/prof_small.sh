# kernel trace of the headline workload in small sub-sweeps: tools/prof_small.sh <rows per sub-sweep> <tag>
B=$1; tag=$2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-strong --batch $B --steps 4 --warmup 1 --other-batches "" --cpu-rows 0 > $out/bench.json 2> $out/log.txt
