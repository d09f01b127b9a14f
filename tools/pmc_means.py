"""Per-kernel means of ONE counter from a rocprofv3 --pmc run, as CSV on stdout.
usage: python tools/pmc_means.py <dir> <COUNTER>"""
import csv
import glob
import sys
from collections import OrderedDict

root, counter = sys.argv[1], sys.argv[2]
acc = OrderedDict()
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
out = csv.writer(sys.stdout)
out.writerow(["Kernel_Name", "Dispatches", "Mean_%s_KB" % counter,
              "Total_%s_KB" % counter])
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    out.writerow([name, len(v), sum(v) / len(v), sum(v)])
