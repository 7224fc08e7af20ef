# Training step: where the main stream idles (gap attribution from the kernel trace)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for bn in frozen; do
rocprofv3 --kernel-trace --output-format csv -d /tmp/r3o_trace_$bn -o run -- python $R/bench.py --train --bn $bn --no-cpu-baseline --no-probe --no-pmc --steps 3 --warmup 3 > $R/gpurun_out/r3o_$bn.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/r3o_trace_$bn/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adamw_multi" in r["Kernel_Name"]]
bursts = []
for i in ad:
    if not bursts or i - bursts[-1][-1] > 200: bursts.append([i])
    else: bursts[-1].append(i)
s0, s1 = bursts[-2][-1] + 1, bursts[-1][-1] + 1
step = rows[s0:s1]
key = "Stream_Id" if "Stream_Id" in step[0] else "Queue_Id"
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
busy = collections.Counter()
for r in step: busy[r[key]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
main = max(busy, key=busy.get)
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
print("$bn step wall %.1f ms, main stream %s busy %.1f ms" % ((t1 - t0) / 1e6, main, busy[main] / 1e6))
m = [r for r in step if r[key] == main]
others = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in step if r[key] != main]
gaps = []
for a, b in zip(m, m[1:]):
    g0, g1 = int(a["End_Timestamp"]), int(b["Start_Timestamp"])
    if g1 - g0 > 3000:
        # time inside the gap covered by kernels of other streams
        cov = 0
        for s, e, n in others:
            lo, hi = max(s, g0), min(e, g1)
            if hi > lo: cov += hi - lo
        gaps.append((g1 - g0, min(cov, g1 - g0), short(a["Kernel_Name"]), short(b["Kernel_Name"]), (g0 - t0) / 1e6))
tot = sum(g[0] for g in gaps); covd = sum(g[1] for g in gaps)
print("gaps > 3 us: %d, total %.2f ms, of which other streams busy %.2f ms" % (len(gaps), tot / 1e6, covd / 1e6))
small = sum(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(m, m[1:]) if 0 < int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) <= 3000)
print("gaps <= 3 us total %.2f ms over %d launches" % (small / 1e6, len(m)))
by_next = collections.Counter(); by_next_n = collections.Counter()
for g in gaps: by_next[g[3]] += g[0]; by_next_n[g[3]] += 1
print("by waiting kernel:", ", ".join("%s %.2f ms (%d)" % (k, v / 1e6, by_next_n[k]) for k, v in by_next.most_common(14)))
print("largest:")
for g in sorted(gaps, reverse=True)[:25]:
    print("  %.0f us (other streams %.0f) after %s -> %s at %.1f ms" % (g[0] / 1e3, g[1] / 1e3, g[2], g[3], g[4]))
# histogram of gap position along the step (10 bins)
bins = [0.0] * 10
for g in gaps: bins[min(9, int(g[4] / ((t1 - t0) / 1e6) * 10))] += g[0] / 1e6
print("gap ms by tenth of the step:", " ".join("%.1f" % b for b in bins))
PY
done
