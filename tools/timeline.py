"""Stream timeline of one step from a `rocprofv3 --kernel-trace` CSV: which queue is busy when, where the main queue idles.

    python tools/timeline.py <run_kernel_trace.csv> [--step-marker KERNEL_SUBSTRING] [--step N] [--gaps 25] [--top 30]

A step = the launches between two consecutive occurrences of the marker kernel (default: the AdamW kernel of the training step,
`adamw_multi_kernel`, first launch of each burst).  Prints, for the chosen step: wall time, busy time per queue, the time
exactly one / two / three queues are busy, the main queue's kernels by total time, and its longest idle gaps with what the other
queues ran meanwhile.  Used to decide what is on the training step's critical path."""
import argparse
import collections
import csv
import re


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    return n.split("(")[0][:70]


ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--step-marker", default="adamw_multi_kernel")
ap.add_argument("--step", type=int, default=-2, help="which step (index into the marker-delimited steps; default: the one before last)")
ap.add_argument("--gaps", type=int, default=25)
ap.add_argument("--top", type=int, default=30)
args = ap.parse_args()

rows = []
for r in csv.DictReader(open(args.trace)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id") or r.get("Stream_Id"), short(r["Kernel_Name"])))
rows.sort()
# step boundaries: first marker launch after a run of >= 50 non-marker launches
marks, since = [], 10 ** 9
for i, (s, e, q, n) in enumerate(rows):
    if args.step_marker in n:
        if since >= 50:
            marks.append(i)
        since = 0
    else:
        since += 1
assert len(marks) >= 3, f"found {len(marks)} steps"
lo, hi = marks[args.step], marks[args.step + 1]
step = rows[lo:hi]
t0, t1 = step[0][0], max(e for s, e, q, n in step)
print(f"step of {len(step)} launches, wall {(t1 - t0) / 1e6:.3f} ms ({len(marks) - 1} steps in the trace)")
byq = collections.defaultdict(list)
for s, e, q, n in step:
    byq[q].append((s, e, n))
main = max(byq, key=lambda q: sum(e - s for s, e, n in byq[q]))
for q, v in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, n in kv[1])):
    print(f"  queue {q}{' (main)' if q == main else ''}: {len(v)} launches, busy {sum(e - s for s, e, n in v) / 1e6:.3f} ms")
# concurrency histogram
ev = sorted([(s, 1) for s, e, q, n in step] + [(e, -1) for s, e, q, n in step])
depth, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[min(depth, 3)] += t - last
    depth, last = depth + d, t
print("  concurrency: " + ", ".join(f"{k}{'+' if k == 3 else ''} kernels {v / 1e6:.3f} ms" for k, v in sorted(hist.items())))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in byq[main]:
    agg[n][0] += 1
    agg[n][1] += e - s
print(f"main queue kernels (top {args.top}):")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
    print(f"  {n:72s} {c:4d} {t / 1e6:8.3f} ms")
for q in byq:
    if q == main:
        continue
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in byq[q]:
        agg[n][0] += 1
        agg[n][1] += e - s
    print(f"queue {q} kernels:")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {n:72s} {c:4d} {t / 1e6:8.3f} ms")
# idle gaps of the main queue
mk = sorted(byq[main])
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(mk, mk[1:]):
    if s1 - e0 > 3000:
        gaps.append((s1 - e0, e0, s1, n0, n1))
print(f"main queue idle: {sum(g[0] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps > 3 us; the longest:")
for g, a, b, n0, n1 in sorted(gaps, reverse=True)[:args.gaps]:
    others = collections.Counter()
    for s, e, q, n in step:
        if q != main and s < b and e > a:
            others[n] += min(e, b) - max(s, a)
    o = ", ".join(f"{n} {t / 1e3:.0f}us" for n, t in others.most_common(3))
    print(f"  {g / 1e3:8.1f} us at +{(a - t0) / 1e6:7.3f} ms  after {n0[:40]:40s} before {n1[:40]:40s} | {o}")
