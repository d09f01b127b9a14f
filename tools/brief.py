"""one-line digest of a bench.py JSON line on stdin"""
import json
import sys
try:
    d = json.loads(sys.stdin.read())
    r = d["roofline"]
    print("value %.4g G/s  ms/step %.2f  kernel %s avg_launch_ms %.3f" % (
        d["value"] / 1e9, d["ms_per_step"], r["kernel"],
        r["avg_launch_ms"] or -1))
except Exception as e:   # noqa: BLE001
    print("no bench line:", e)
