"""Every dispatch of ONE pass (sweep) of a rocprofv3 --kernel-trace run, with
the gaps: python tools/pass_timeline.py <dir> <anchor> <launches-per-pass>
prints the last complete pass that starts at an anchor launch: per kernel
name the count, the summed duration, and every gap above 3 us."""
import csv
import glob
import sys
from collections import OrderedDict

root, anchor, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a = idx[-2 * per - 1]
b = idx[-per - 1]
t0 = int(rows[a]["Start_Timestamp"])
agg = OrderedDict()
prev_end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:60]
    c = agg.setdefault(name, [0, 0.0])
    c[0] += 1
    c[1] += (e - s) / 1e3
    if prev_end is not None and (s - prev_end) / 1e3 > 3.0:
        print("gap %7.1f us before %s at %9.1f" % ((s - prev_end) / 1e3, name,
                                                   (s - t0) / 1e3))
    prev_end = max(e, prev_end or e)
for name, (n, d) in agg.items():
    print("%4d x %-60s %9.1f us" % (n, name, d))
print("pass: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
