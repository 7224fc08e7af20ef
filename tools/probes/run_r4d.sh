#!/bin/bash
# round 4: per-shape GEMM times inside the step, LayerNorm fold on / off (inline schedule figures)
R=$GRAFT_REPO_ROOT; cd $R
for v in 1 0; do
  MSCLIP_LN_FOLD=$v python bench.py --no-cpu-baseline --no-pmc --shapes "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('fold=$v', r['value'], r['ms_per_step'])
for s in r['roofline']['shapes'][:8]:
    print('   ', s['M'], s['N'], s['K'], 'act', s['act'], 'resid', s['resid'], 'n', s['launches_per_step'], s['avg_us'], 'us', s['tflops'], 'TF')
"
done
