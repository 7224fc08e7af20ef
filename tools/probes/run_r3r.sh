#!/bin/bash
# Training step: HIP API calls of the host thread (which of them block?)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --stats --output-format csv -d /tmp/r3r -o run -- python $R/bench.py --train --bn frozen --no-cpu-baseline --no-probe --no-pmc --steps 6 --warmup 3 > $R/gpurun_out/r3r.log 2>&1
ls /tmp/r3r/* | head
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/r3r/**/*hip_api_stats.csv", recursive=True) or glob.glob("/tmp/r3r/**/*hip*stats*.csv", recursive=True)
print(f)
for r in list(csv.DictReader(open(f[0])))[:25]:
    print(r)
t = glob.glob("/tmp/r3r/**/*hip_api_trace.csv", recursive=True)
if t:
    rows = list(csv.DictReader(open(t[0])))
    print(len(rows), "api calls", list(rows[0].keys()))
    # longest individual calls that are not launches
    rows.sort(key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)
    for r in rows[:40]:
        print(r["Function"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
