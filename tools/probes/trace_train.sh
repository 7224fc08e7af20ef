# rocprofv3 kernel trace of the training step + timeline + a per-launch dump of one step (gpurun_out/<tag>_step_launches.txt)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out; TAG=${1:-r5t}; BN=${2:-frozen}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc --no-hbm-kernels --train --bn $BN --steps 8 --warmup 3 > $O/${TAG}_prof.log 2>&1
T=$(find $O/${TAG}_prof -name "*kernel_trace.csv" | head -1)
cd $R; python tools/timeline.py $T --top 45 --gaps 14 > $O/${TAG}_timeline.txt 2>&1
python - "$T" > $O/${TAG}_step_launches.txt <<'PY'
import csv,sys,re
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]).replace("void ","").split("(")[0][:64]
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r.get("Queue_Id"),n))
rows.sort()
marks=[];since=10**9
for i,(s,e,q,n) in enumerate(rows):
    if "adamw_multi" in n:
        if since>=50: marks.append(i)
        since=0
    else: since+=1
lo,hi=marks[-3],marks[-2]
t0=rows[lo][0]
for s,e,q,n in rows[lo:hi]:
    print(f"{(s-t0)/1e3:10.1f} {(e-s)/1e3:8.1f} q{q} {n}")
PY
find $O/${TAG}_prof -name "*kernel_trace.csv" -delete; find $O/${TAG}_prof -name "*agent_info.csv" -delete
head -5 $O/${TAG}_timeline.txt
