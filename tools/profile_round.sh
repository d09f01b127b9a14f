#!/bin/bash
# The round's evidence, on the GPU box: tools/profile_round.sh <out-dir under gpurun_out>
# (kernel trace + stats, FETCH_SIZE / WRITE_SIZE in separate passes, the bench
# line of record with the CPU baseline, the other configurations)
set -u
out=gpurun_out/$1
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp && cd $repo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --cpu-rows 0 > $out/bench_under_rocprof.json 2> $out/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-rows 0 > /dev/null 2> $out/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --cpu-rows 0 > /dev/null 2> $out/write.log
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
python tools/pmc_means.py $out/fetch FETCH_SIZE > $out/pmc_fetch_size.csv
python tools/pmc_means.py $out/write WRITE_SIZE > $out/pmc_write_size.csv
rm -rf $out/trace $out/fetch $out/write
python bench.py 2> $out/bench.log | tail -1 > $out/bench.json
: > $out/bench_other_configs.jsonl
# (3 timed sweeps after 1: on pure-noise reals the groups of the NormalInverseChiSq
# configurations merge sweep by sweep, and the per-group ordered replay of a
# group of 10^6 rows is one dependent chain -- the numbers are for K ~ 1024)
for c in gp_nich nich gp bb mixed; do
  python bench.py --cpu-rows 0 --steps 3 --warmup 1 --config $c 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
done
python bench.py --cpu-rows 0 --steps 3 --warmup 2 --config dpd --groups 8192 --dim 10000 2>/dev/null | tail -1 >> $out/bench_other_configs.jsonl
python bench.py --cpu-rows 0 --force-collective 2>/dev/null | tail -1 > $out/bench_collective_one_rank.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3 -- python3 bench.py --cpu-rows 0 --steps 3 --warmup 1 --config gp_nich > /dev/null 2>&1
cp $(ls $out/c3/*/*kernel_stats.csv | head -1) $out/kernel_stats_c3_gp_nich.csv; rm -rf $out/c3
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -- python3 bench.py --cpu-rows 0 --steps 3 --warmup 2 --config dpd --groups 8192 --dim 10000 > /dev/null 2>&1
cp $(ls $out/c5/*/*kernel_stats.csv | head -1) $out/kernel_stats_c5_dpd.csv; rm -rf $out/c5
ls -la $out
