"""Per-launch counter medians of named kernels from rocprofv3 --pmc runs, as
the JSON bench.py reads (profiles/r2_counters.json).

usage: python tools/counters.py OUT.json NAME=SUBSTR:ROWS_PER_LAUNCH[:GRID_MIN] ... -- DIR ...
  NAME     the kernel as bench.py calls it, e.g. "k_vs_sample<dd>"
  SUBSTR   what its dispatches' Kernel_Name contains, e.g. "k_vs_sample<0>"
  DIR      rocprofv3 output directories (each from ONE --pmc pass)
Only dispatches of at least GRID_MIN threads count (the steady-state launches
of the bench, not the small ones of a test).  FETCH_SIZE / WRITE_SIZE are in
KB; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_BUSY_CYCLES count
quad-cycles (MI355X_MICROARCH.md, cycle constants)."""
import csv
import glob
import hashlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "distributions_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) and name != "ref_tables.h":
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def main():
    out_path = sys.argv[1]
    sep = sys.argv.index("--")
    specs = {}
    for a in sys.argv[2:sep]:
        name, rest = a.split("=", 1)
        parts = rest.split(":")
        specs[name] = (parts[0], int(parts[1]),
                       int(parts[2]) if len(parts) > 2 else 0)
    vals = {name: {} for name in specs}      # name -> counter -> [values]
    durs = {name: [] for name in specs}
    for d in sys.argv[sep + 1:]:
        for path in glob.glob(d + "/**/*counter_collection.csv",
                              recursive=True):
            for row in csv.DictReader(open(path)):
                for name, (sub, _, grid_min) in specs.items():
                    if sub in row["Kernel_Name"] and int(
                            row["Grid_Size"]) >= grid_min:
                        vals[name].setdefault(row["Counter_Name"], []).append(
                            float(row["Counter_Value"]))
                        durs[name].append(int(row["End_Timestamp"])
                                          - int(row["Start_Timestamp"]))
    kernels = {}
    for name, (sub, rows, _) in specs.items():
        c = {k: statistics.median(v) for k, v in vals[name].items()}
        if not c:
            continue
        rec = {"match": sub, "rows_per_launch": rows,
               "dispatches": {k: len(v) for k, v in vals[name].items()},
               "median": c,
               "duration_ns_under_pmc": statistics.median(durs[name])}
        if "FETCH_SIZE" in c:
            rec["fetch_bytes"] = c["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in c:
            rec["write_bytes"] = c["WRITE_SIZE"] * 1024.0
        if "SQ_ACTIVE_INST_VALU" in c:
            rec["valu_busy_cycles"] = c["SQ_ACTIVE_INST_VALU"] * 4.0
        if "SQ_INSTS_VALU" in c:
            rec["valu_instructions"] = c["SQ_INSTS_VALU"]
        kernels[name] = rec
    json.dump({"source_hash": source_hash(), "kernels": kernels,
               "units": "FETCH_SIZE/WRITE_SIZE KB -> bytes; "
                        "valu_busy_cycles = SQ_ACTIVE_INST_VALU x 4 "
                        "(quad-cycles -> cycles), summed over the chip's "
                        "1024 SIMDs"},
              open(out_path, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
