#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "captured_step or adamw_multi" 2>&1 | tail -25
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; }
timeout 600 python bench.py --train --bn frozen --graph --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/tmp/e1 | tail -1 | line graph_frozen || tail -20 /tmp/e1
timeout 600 python bench.py --train --bn batch --graph --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/tmp/e2 | tail -1 | line graph_batch || tail -20 /tmp/e2
timeout 600 python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 | line eager_frozen
