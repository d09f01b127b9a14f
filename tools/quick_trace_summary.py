"""one configuration of tools/quick_trace.sh: the bench value and the
value-sorted kernels' average durations"""
import csv
import json
import sys

name = sys.argv[1]
line = open("gpurun_out/quick_bench_%s.json" % name).read().strip().splitlines()[-1]
print("==", name, "%.4g row-updates/s (under rocprofv3)" % json.loads(line)["value"])
for r in csv.DictReader(open("gpurun_out/quick_stats_%s.csv" % name)):
    n = r["Name"]
    if "k_vs_" in n and int(r["Calls"]) > 20:
        print("  %-28s calls %5s avg %7.1f us" % (
            n.split("(")[0].replace("void dist::", "")[:28], r["Calls"],
            float(r["AverageNs"]) / 1e3))
