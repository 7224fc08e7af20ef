#!/bin/bash
# single-query last-block attention: tests, A/B; per-launch GEMM durations overlapped vs inline
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "last_block or golden or zero_shot or graph" 2>&1 | tail -3
for i in 1 2; do
  MSCLIP_LAST_BLOCK_ALL_QUERIES=1 python bench.py --no-cpu-baseline --no-pmc --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('allq', d['value'], d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-pmc --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lastq', d['value'], d['ms_per_step'])"
done
MSCLIP_LAST_BLOCK_ALL_QUERIES=1 python bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b16 allq', d['value'], d['ms_per_step'])"
python bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b16 lastq', d['value'], d['ms_per_step'])"
for mode in ov inline; do
  rm -rf /tmp/tr_$mode
  if [ $mode = inline ]; then export MSCLIP_CONV_SIDE_STREAM=0; else unset MSCLIP_CONV_SIDE_STREAM; fi
  rocprofv3 --kernel-trace -d /tmp/tr_$mode -o run --output-format csv -- python bench.py --no-cpu-baseline --no-pmc --no-probe --steps 10 --warmup 5 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/tr_$mode/**/*kernel_trace.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "gemm_pp_kernel<0, false" in n:
        d["pp0"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
v=sorted(d["pp0"]); n=len(v)
print("$mode", n, "min %.0f p10 %.0f p25 %.0f med %.0f p75 %.0f p90 %.0f max %.0f mean %.1f"%(v[0],v[n//10],v[n//4],v[n//2],v[3*n//4],v[int(n*.9)],v[-1],sum(v)/n))
# last full step: per-launch list
last=d["pp0"][-50:]
print("$mode last step:", " ".join("%.0f"%x for x in last))
PY
done
