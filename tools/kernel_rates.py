"""Per-kernel HBM traffic / rate table from one bench configuration's rocprofv3 passes:
    python tools/kernel_rates.py <stats_dir> <pmc_prefix> <steps_profiled> "<title>"  > profiles/rNN_kernel_hbm_rates.md
<stats_dir>/run_kernel_stats.csv from `rocprofv3 --kernel-trace --stats`; <pmc_prefix>_{FETCH_SIZE,WRITE_SIZE}/ from
tools/pmc_bench.sh.  FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported."""
import collections
import csv
import re
import sys

stats_dir, prefix, steps, title = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
st = {}
for r in csv.DictReader(open(f"{stats_dir}/run_kernel_stats.csv")):
    st[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
pm = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{prefix}_{c}/run_counter_collection.csv")):
        if r["Counter_Name"] == c:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    pm[c] = {k: sum(v) / len(v) for k, v in agg.items()}
print(f"# {title}\n")
print("PMC `FETCH_SIZE` (x2: the gfx950 correction of MI355X_MICROARCH.md) and `WRITE_SIZE` per launch from the two `--pmc`\n"
      "passes of `tools/pmc_bench.sh` (bytes that crossed the L2 <-> fabric boundary: HBM or the 256 MB MALL), average launch\n"
      "time and share of the step from the `--kernel-trace --stats` pass of the same command.\n")
print("| kernel | launches/step | avg µs | % of step | MB read | MB written | TB/s |\n|---|---|---|---|---|---|---|")
for k, (calls, us, pct) in sorted(st.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if k.startswith(("at::", "__amd", "rocblas")) or pct < 0.1:
        continue
    rd, wr = 2 * pm["FETCH_SIZE"].get(k, 0.0) * 1024 / 1e6, pm["WRITE_SIZE"].get(k, 0.0) * 1024 / 1e6
    print(f"| `{k}` | {calls / steps:g} | {us:.1f} | {pct:.2f} | {rd:.0f} | {wr:.0f} | {(rd + wr) / us:.2f} |")
