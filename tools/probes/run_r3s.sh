#!/bin/bash
# Training step: which memory copies block the host, and between which kernels
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/r3s -o run -- python $R/bench.py --train --bn ${1:-frozen} --no-cpu-baseline --no-probe --no-pmc --steps 2 --warmup 3 > $R/gpurun_out/r3s.log 2>&1
ls /tmp/r3s/
python - <<'PY'
import csv, glob, collections, bisect
api = list(csv.DictReader(open(glob.glob("/tmp/r3s/*hip_api_trace.csv")[0])))
ker = list(csv.DictReader(open(glob.glob("/tmp/r3s/*kernel_trace.csv")[0])))
mc = glob.glob("/tmp/r3s/*memory_copy_trace.csv")
if mc:
    rows = list(csv.DictReader(open(mc[0])))
    print("memory copies:", len(rows), list(rows[0].keys()))
    c = collections.Counter((r.get("Direction"), r.get("Size") if "Size" in r else None) for r in rows)
    for k, v in c.most_common(25): print("  ", k, v)
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
api.sort(key=lambda r: int(r["Start_Timestamp"]))
by_corr = {r["Correlation_Id"]: short(r["Kernel_Name"]) for r in ker}
# last 40 % of the trace = the timed steps; list each blocking memcpy with the launches around it
launches = [(int(r["Start_Timestamp"]), by_corr.get(r["Correlation_Id"], "?")) for r in api if r["Function"] == "hipLaunchKernel"]
ts = [t for t, _ in launches]
mem = [r for r in api if r["Function"] == "hipMemcpyWithStream"]
print("hipMemcpyWithStream calls:", len(mem))
ctx = collections.Counter()
for r in mem[len(mem) // 2:]:
    i = bisect.bisect_left(ts, int(r["Start_Timestamp"]))
    before = launches[i - 1][1] if i else "-"
    after = launches[i][1] if i < len(launches) else "-"
    ctx[(before, after)] += 1
for k, v in ctx.most_common(40): print(v, k)
PY
