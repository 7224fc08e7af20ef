R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
: > $O/r5c_ab.txt
for i in 1 2 3; do
  for early in 0 1; do
    MSCLIP_BRANCH_EARLY=$early python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('early=$early', r['ms_per_step'], r['value'])" >> $O/r5c_ab.txt
  done
done
cat $O/r5c_ab.txt
python bench.py --shapes --no-cpu-baseline --no-pmc --no-hbm-kernels 2>/dev/null | tail -1 > $O/r5c_shapes.json
python -c "
import json; r=json.load(open('$O/r5c_shapes.json')); print(json.dumps(r.get('roofline',{}).get('shapes', r.get('shapes')), indent=0)[:3000])"
