#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "gradients or cast_with or training_loop" 2>&1 | tail -2
for i in 1 2; do for bn in frozen batch; do echo -n "$bn: "; python bench.py --train --bn $bn --no-cpu-baseline --no-pmc --no-probe --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done; done
echo -n "b16 batch: "; python bench.py --train --bn batch --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-pmc --no-probe --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
