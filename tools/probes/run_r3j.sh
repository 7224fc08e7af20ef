R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
for a in 0 1; do for b in 0 1; do
  echo -n "conv_side=$a adapter_lane=$b frozen: "; MSCLIP_TRAIN_CONV_SIDE=$a MSCLIP_TRAIN_ADAPTER_LANE=$b python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
done; done; done
