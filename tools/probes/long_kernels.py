"""The launches of one traced training step that take longer than a threshold, in issue order, with their grids."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "").split("(")[0][:60]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "stem_dual_mfma" in r["Kernel_Name"]]
lo, hi = marks[-2], marks[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d >= thr and "gemm_pp_kernel<0" not in r["Kernel_Name"]:
        print(f"+{(int(r['Start_Timestamp']) - t0) / 1e6:7.2f} ms q{r['Queue_Id']} {short(r['Kernel_Name']):60s} {d:8.1f} us grid {r.get('Grid_Size_X', r.get('Grid_Size'))}x{r.get('Grid_Size_Y', '')} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size'))}")
