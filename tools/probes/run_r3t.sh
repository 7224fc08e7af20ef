#!/bin/bash
# Training step: for the largest idle gaps of the main stream -- was the next kernel already launched by the host?
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d /tmp/r3t -o run -- python $R/bench.py --train --bn ${1:-frozen} --no-cpu-baseline --no-probe --no-pmc --steps 3 --warmup 3 > $R/gpurun_out/r3t.log 2>&1
python - <<'PY'
import csv, glob, collections, bisect
api = list(csv.DictReader(open(glob.glob("/tmp/r3t/*hip_api_trace.csv")[0])))
ker = list(csv.DictReader(open(glob.glob("/tmp/r3t/*kernel_trace.csv")[0])))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
launch_t = {r["Correlation_Id"]: (int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in api if r["Function"] == "hipLaunchKernel"}
ker.sort(key=lambda r: int(r["Start_Timestamp"]))
key = "Stream_Id" if "Stream_Id" in ker[0] else "Queue_Id"
ad = [i for i, r in enumerate(ker) if "adamw_multi" in r["Kernel_Name"]]
bursts = []
for i in ad:
    if not bursts or i - bursts[-1][-1] > 200: bursts.append([i])
    else: bursts[-1].append(i)
step = ker[bursts[-2][-1] + 1: bursts[-1][-1] + 1]
busy = collections.Counter()
for r in step: busy[r[key]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
main = max(busy, key=busy.get)
m = [r for r in step if r[key] == main]
t0 = int(step[0]["Start_Timestamp"])
gaps = []
for a, b in zip(m, m[1:]):
    g0, g1 = int(a["End_Timestamp"]), int(b["Start_Timestamp"])
    if g1 - g0 > 20000: gaps.append((g1 - g0, g0, g1, a, b))
gaps.sort(reverse=True, key=lambda g: g[0])
slow = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in api if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 50000]
host_bound = gpu_wait = 0
for g, g0, g1, a, b in gaps:
    lt = launch_t.get(b["Correlation_Id"])
    if lt is None: continue
    if lt[0] >= g0 - 5000: host_bound += g
    else: gpu_wait += g
print("main-stream gaps > 20 us: %d; next kernel launched by the host AFTER the gap began: %.2f ms; launched earlier (GPU-side wait): %.2f ms" % (len(gaps), host_bound / 1e6, gpu_wait / 1e6))
for g, g0, g1, a, b in gaps[:12]:
    lt = launch_t.get(b["Correlation_Id"])
    print("gap %.0f us at %.1f ms: %s -> %s; host launched the latter at gap_start %+.0f us (call took %.0f us)" % (
        g / 1e3, (g0 - t0) / 1e6, short(a["Kernel_Name"]), short(b["Kernel_Name"]), (lt[0] - g0) / 1e3, (lt[1] - lt[0]) / 1e3))
    for s, e, f in slow:
        if e > g0 - 3_000_000 and s < g1:
            print("      host call %s %.0f us, from gap_start %+.0f us" % (f, (e - s) / 1e3, (s - g0) / 1e3))
PY
