"""Summarise rocprofv3 --pmc CSV output for kernels matching a substring: python tools/pmc_summary.py <dir-prefix> [substr]"""
import collections
import csv
import glob
import sys

prefix = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "gemm"
for d in sorted(glob.glob(prefix + "*")):
    f = d + "/run_counter_collection.csv"
    try:
        rows = list(csv.DictReader(open(f)))
    except Exception:
        continue
    agg = collections.defaultdict(list)
    for r in rows:
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for c, v in sorted(agg.items()):
        vals = [x[0] for x in v]
        dur = [x[1] for x in v]
        print(f"{d.split('/')[-1]:22s} {c:32s} n={len(v):3d} mean={sum(vals)/len(vals):16.1f} dur_us={sum(dur)/len(dur)/1e3:9.1f}")
