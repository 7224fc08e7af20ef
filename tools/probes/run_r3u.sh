#!/bin/bash
# Training step: host time inside hipLaunchKernel by kernel / stream; what precedes the slow ones
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d /tmp/r3u -o run -- python $R/bench.py --train --bn ${1:-frozen} --no-cpu-baseline --no-probe --no-pmc --steps 3 --warmup 3 > $R/gpurun_out/r3u.log 2>&1
python - <<'PY'
import csv, glob, collections
api = list(csv.DictReader(open(glob.glob("/tmp/r3u/*hip_api_trace.csv")[0])))
ker = list(csv.DictReader(open(glob.glob("/tmp/r3u/*kernel_trace.csv")[0])))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
key = "Stream_Id" if "Stream_Id" in ker[0] else "Queue_Id"
kinfo = {r["Correlation_Id"]: (short(r["Kernel_Name"]), r[key]) for r in ker}
api.sort(key=lambda r: int(r["Start_Timestamp"]))
# main thread only, last 40 % of the calls (the timed steps)
tid = collections.Counter(r["Thread_Id"] for r in api).most_common(1)[0][0]
api = [r for r in api if r["Thread_Id"] == tid]
api = api[int(len(api) * 0.6):]
span = (int(api[-1]["End_Timestamp"]) - int(api[0]["Start_Timestamp"])) / 1e6
tot = collections.Counter(); cnt = collections.Counter()
for r in api:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[r["Function"]] += d; cnt[r["Function"]] += 1
print("window %.1f ms of host time; inside HIP calls:" % span, ", ".join("%s %.1f ms (%d)" % (k, v / 1e6, cnt[k]) for k, v in tot.most_common(8)))
byk = collections.defaultdict(list)
prev = None
after = collections.Counter(); after_n = collections.Counter()
for r in api:
    if r["Function"] == "hipLaunchKernel":
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name, st = kinfo.get(r["Correlation_Id"], ("?", "?"))
        byk[(name, st)].append(d)
        after[prev] += d; after_n[prev] += 1
    prev = r["Function"]
print("hipLaunchKernel by (kernel, stream): total ms, calls, median us, p90 us")
for k, v in sorted(byk.items(), key=lambda kv: -sum(kv[1]))[:30]:
    v = sorted(v)
    print("  %-42s st %-3s %7.2f ms %5d  med %5.1f  p90 %5.1f" % (k[0], k[1], sum(v) / 1e3, len(v), v[len(v) // 2], v[int(len(v) * .9)]))
print("hipLaunchKernel time by the HIP call right before it:")
for k, v in after.most_common(8): print("  after %-28s %7.2f ms over %5d launches (%.1f us each)" % (k, v / 1e3, after_n[k], v / after_n[k]))
PY
