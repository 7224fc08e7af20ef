#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
t() { b=$1; bn=$2; g=$3; shift 3; echo "== batch $b bn $bn $g $*"; env "$@" timeout 400 python bench.py --batch $b --train --bn $bn $g --no-cpu-baseline --no-pmc --no-probe --steps 8 --warmup 3 > /tmp/o1 2> /tmp/e1; echo rc=$?; python -c "
import json,sys
try:
    d=json.loads(open('/tmp/o1').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('launch'))
except Exception as e: print('no json', e)
"; }
t 48 frozen --graph A=1
t 512 frozen --graph A=1
t 512 frozen --no-pmc A=1
t 512 batch --graph A=1
t 512 batch --no-pmc A=1
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k 'captured_step or gradients or splitk' 2>&1 | tail -4
