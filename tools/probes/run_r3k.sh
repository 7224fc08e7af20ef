R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
MSCLIP_CONV_SIDE_STREAM=$v rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3k_trace -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc --steps 6 --warmup 3 > $R/gpurun_out/r3k.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/r3k_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "loss_from_partials" in r["Kernel_Name"]]
s0, s1 = ends[-3] + 1, ends[-2] + 1
step = rows[s0:s1]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
print("side streams =", $v, "step wall ms", (t1 - t0) / 1e6, "launches", len(step))
busy = collections.defaultdict(float); cnt = collections.Counter()
for r in step:
    busy[r["Stream_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; cnt[r["Stream_Id"]] += 1
print("  busy per stream:", {k: (round(v, 2), cnt[k]) for k, v in busy.items()})
# gaps on the busiest stream
main = max(busy, key=busy.get)
m = [r for r in step if r["Stream_Id"] == main]
gaps = [(int(m[i + 1]["Start_Timestamp"]) - int(m[i]["End_Timestamp"])) / 1e3 for i in range(len(m) - 1)]
print("  main stream: sum of gaps %.2f ms, median gap %.2f us, gaps > 10 us: %d (%.2f ms)" % (sum(g for g in gaps if g > 0) / 1e3, sorted(gaps)[len(gaps) // 2], sum(1 for g in gaps if g > 10), sum(g for g in gaps if g > 10) / 1e3))
big = sorted([(g, m[i]["Kernel_Name"][:60], m[i + 1]["Kernel_Name"][:60]) for i, g in enumerate(gaps)], reverse=True)[:8]
for g, a, b in big: print("   gap %.1f us after %s before %s" % (g, a, b))
PY
rm -rf $R/gpurun_out/r3k_trace
done
