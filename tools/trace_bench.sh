# kernel-trace averages of the headline bench: tools/trace_bench.sh <tag> [bench args]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-strong --other-batches "" --cpu-rows 0 "$@" > $out/bench.json 2> $out/log.txt
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
