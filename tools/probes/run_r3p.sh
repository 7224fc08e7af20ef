#!/bin/bash
# optimizer step without the host-side re-pack: tests, host cost, step time
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "adamw or optimizer or training_loop or checkpoint or odd_batches or train_mode_batchnorm" 2>&1 | tail -5
python tools/probes/pack_cost.py frozen 2>/dev/null
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 | line train_frozen
python bench.py --train --bn batch --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 | line train_batch
