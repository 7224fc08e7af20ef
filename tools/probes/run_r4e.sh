#!/bin/bash
# round 4: training step with the token-major weight-gradient GEMM (no operand transposes) on / off, same box, alternating
R=$GRAFT_REPO_ROOT; cd $R
for pass in 1 2; do
for v in 1 0; do
  for bn in frozen batch; do
    MSCLIP_WGRAD_TN=$v python bench.py --train --bn $bn --no-cpu-baseline --no-pmc --no-probe --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('tn=$v bn=$bn', r['value'], r['ms_per_step'], r['loss'])"
  done
done
done
