R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for bn in frozen batch; do
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3i_trace_$bn -o run -- python $R/bench.py --train --bn $bn --no-cpu-baseline --no-probe --no-pmc --steps 3 --warmup 3 > $R/gpurun_out/r3i_$bn.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/r3i_trace_$bn/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("$bn", "columns", list(rows[0].keys()))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last third = one steady step (3 warmup + 3 timed): take kernels of the last step by splitting on adamw launches
ad = [i for i, r in enumerate(rows) if "adamw_multi" in r["Kernel_Name"]]
# steps end at the last adamw launch of a burst: find bursts
bursts = []
for i in ad:
    if not bursts or i - bursts[-1][-1] > 200: bursts.append([i])
    else: bursts[-1].append(i)
s0, s1 = bursts[-2][-1] + 1, bursts[-1][-1] + 1
step = rows[s0:s1]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
print("step wall ms", (t1 - t0) / 1e6, "launches", len(step))
key = "Stream_Id" if "Stream_Id" in step[0] else "Queue_Id"
busy = collections.defaultdict(float); cnt = collections.Counter(); names = collections.defaultdict(lambda: collections.defaultdict(float))
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    busy[r[key]] += d; cnt[r[key]] += 1
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    names[r[key]][n] += d
for q in sorted(busy, key=lambda q: -busy[q]):
    print(f"  {key} {q}: busy {busy[q]:.1f} ms, {cnt[q]} launches; top:", ", ".join(f"{n} {v:.1f}" for n, v in sorted(names[q].items(), key=lambda kv: -kv[1])[:14]))
PY
rm -rf $R/gpurun_out/r3i_trace_$bn
done
