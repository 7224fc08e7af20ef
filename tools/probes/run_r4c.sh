#!/bin/bash
# round 4: LayerNorm fold on / off, same box, alternating (C2 default; pass extra bench args)
R=$GRAFT_REPO_ROOT; cd $R
for pass in 1 2; do
for v in 1 0; do
  MSCLIP_LN_FOLD=$v python bench.py --no-cpu-baseline --no-pmc "$@" 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('fold=$v', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['avg_launch_us'], r['roofline'].get('timed_region_overlapped',{}).get('avg_launch_us'), r['loss'])"
done
done
