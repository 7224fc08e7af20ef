"""HBM traffic per launch from the two rocprofv3 --pmc passes of tools/pmc_bench.sh (FETCH_SIZE, WRITE_SIZE; KiB units).
    python tools/pmc_traffic.py gpurun_out/pmcb_<tag> gemm_pp_kernel > profiles/r01_bench_hbm_traffic.json
FETCH_SIZE is doubled: gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B (MI355X_MICROARCH.md, HBM
section); WRITE_SIZE is taken as reported."""
import collections
import csv
import json
import re
import sys

prefix, dominant = sys.argv[1], sys.argv[2]
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")[:48]
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{prefix}_{c}/run_counter_collection.csv")):
        if r["Counter_Name"] == c:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    per[c] = {k: {"launches": len(v), "mean_kb": sum(v) / len(v)} for k, v in agg.items() if sum(v) / len(v) > 20000}
def pick(table):   # exact name, else the kernel whose name starts with it and has the most launches (template suffixes)
    if dominant in table:
        return table[dominant]
    cands = [v for k, v in table.items() if k.startswith(dominant)]
    return max(cands, key=lambda v: v["launches"])

f = pick(per["FETCH_SIZE"])["mean_kb"]
w = pick(per["WRITE_SIZE"])["mean_kb"]
print(json.dumps({
    "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe (separate passes)",
    "kernel": dominant, "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w,
    "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests of wide coalesced reads at 64 B; MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated, taken as is",
    "hbm_bytes_per_launch": (2 * f + w) * 1024, "per_kernel": per}, indent=1))
