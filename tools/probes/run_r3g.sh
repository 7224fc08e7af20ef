R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for t in _r2tree .; do
  cd $R/$t
  for bn in batch frozen; do
    echo -n "$t bn=$bn: "; python bench.py --train --bn $bn --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
  done
done; done
