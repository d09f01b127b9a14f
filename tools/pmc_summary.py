"""Mean of every counter per kernel from rocprofv3 counter_collection.csv files.
usage: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        if want in name:
            acc[name.split("(")[0][:60]][row["Counter_Name"]].append(
                float(row["Counter_Value"]))
for kernel, counters in sorted(acc.items()):
    print(kernel)
    for c, v in sorted(counters.items()):
        print("   %-28s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
