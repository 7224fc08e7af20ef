"""What is the host doing while the main queue idles?  kernel trace + HIP API trace of the same run (rocprofv3 --hip-trace
--kernel-trace): for the longest idle gaps of the busiest queue, the HIP API calls in flight and the longest calls nearby."""
import collections
import csv
import re
import sys

ktrace, atrace = sys.argv[1], sys.argv[2]
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "").split("(")[0][:50]
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short(r["Kernel_Name"])) for r in csv.DictReader(open(ktrace))]
A = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(atrace))]
K.sort(); A.sort()
byq = collections.defaultdict(list)
for k in K:
    byq[k[2]].append(k)
main = max(byq, key=lambda q: sum(e - s for s, e, _, _ in byq[q]))
mk = byq[main]
t_lo = mk[len(mk) // 3][0]                        # skip the warm-up third
gaps = sorted(((b[0] - a[1], a, b) for a, b in zip(mk, mk[1:]) if a[1] > t_lo), reverse=True)[:12]
print("API calls by total time:")
agg = collections.Counter(); cnt = collections.Counter()
for s, e, f in A:
    if s > t_lo:
        agg[f] += e - s; cnt[f] += 1
for f, t in agg.most_common(12):
    print(f"  {f:40s} {cnt[f]:7d} calls {t / 1e6:9.3f} ms")
for g, a, b in gaps:
    print(f"gap {g / 1e3:8.1f} us after {a[3]} before {b[3]}")
    inflight = [(s, e, f) for s, e, f in A if s < b[0] and e > a[1]]
    inflight.sort(key=lambda x: x[0] - x[1])
    for s, e, f in inflight[:5]:
        print(f"     {f:36s} {(e - s) / 1e3:9.1f} us  starts {(s - a[1]) / 1e3:+9.1f} us, ends {(e - a[1]) / 1e3:+9.1f} us rel. gap start")
    # the launch call of kernel b: the hipLaunchKernel / hipModuleLaunchKernel that ends closest before b starts
    launches = [(s, e, f) for s, e, f in A if "Launch" in f and e <= b[0] + 2000 and e > a[1] - 5_000_000]
    if launches:
        s, e, f = launches[-1]
        print(f"     last launch call before the next kernel: {f} at {(s - a[1]) / 1e3:+.1f} us rel. gap start")

# every HIP API call longer than 150 us in the second half of the trace, with the kernel it launches (correlation id) and its
# position relative to the next first-forward kernel (stem_dual_mfma_kernel)
corr = {}
for r in csv.DictReader(open(ktrace)):
    corr[r["Correlation_Id"]] = short(r["Kernel_Name"])
stems = [k[0] for k in K if "stem_dual_mfma" in k[3]]
print("\nAPI calls > 150 us (second half of the trace):")
rows = list(csv.DictReader(open(atrace)))
t_half = int(rows[len(rows) // 2]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > t_half and e - s > 150_000:
        nxt = min((x for x in stems if x > s), default=None)
        rel = f"{(s - nxt) / 1e6:+8.2f} ms rel. next stem" if nxt else ""
        print(f"  {r['Function']:28s} {(e - s) / 1e3:9.1f} us  {corr.get(r['Correlation_Id'], ''):50s} {rel}")
