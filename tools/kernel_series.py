"""Per-kernel duration over a run, from a rocprofv3 --kernel-trace directory:
python tools/kernel_series.py <dir> <kernel-substring> [buckets]
prints count, mean, min, median, p90, max (us) and the mean per bucket of the
run (first .. last tenth), to see what drifts as the chain goes on."""
import csv
import glob
import sys

root, name = sys.argv[1], sys.argv[2]
buckets = int(sys.argv[3]) if len(sys.argv) > 3 else 10
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]),
              key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
if not d:
    sys.exit("no such kernel")
s = sorted(d)
print("%s: n %d mean %.1f min %.1f median %.1f p90 %.1f max %.1f" % (
    name, len(d), sum(d) / len(d), s[0], s[len(s) // 2], s[int(len(s) * 0.9)],
    s[-1]))
step = max(1, len(d) // buckets)
print("  by tenth of the run:", " ".join(
    "%.1f" % (sum(d[i:i + step]) / len(d[i:i + step]))
    for i in range(0, len(d), step)))
if len(sys.argv) > 4:   # the last launches one by one
    print("  last %s:" % sys.argv[4], " ".join("%.1f" % x for x in d[-int(sys.argv[4]):]))
