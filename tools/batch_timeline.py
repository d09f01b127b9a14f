"""One sub-sweep's launches from a rocprofv3 --kernel-trace run, in order:
python tools/batch_timeline.py <dir> <anchor-kernel-substring> [which]
prints every dispatch between the `which`-th last and the next occurrence of
the anchor kernel: start offset, duration, gap to the previous end (us)."""
import csv
import glob
import sys

root, anchor = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-which - 1], idx[-which]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    print("%9.1f  dur %8.1f  gap %6.1f  %s" % (
        (s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"].split("(")[0][:70]))
    prev_end = max(e, prev_end or e)
print("sub-sweep: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
