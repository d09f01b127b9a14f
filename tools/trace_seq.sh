# the sequence of kernels of one steady-state sub-sweep: tools/trace_seq.sh <tag> <kernel that starts a sub-sweep> [bench args]
tag=$1; first=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-strong --other-batches "" --cpu-rows 0 --steps 1 --warmup 1 "$@" > /dev/null 2> $out/log.txt
python3 - $out $first <<'PY'
import csv, glob, sys
out, first = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]["Start_Timestamp"])
with open(out + "/sequence.txt", "w") as w:
    for r in rows[a:b]:
        w.write("%8.1f us  +%7.1f  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                r["Kernel_Name"][:90]))
    w.write("%8.1f us  (next sub-sweep starts)\n" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
rm -rf $out/trace
