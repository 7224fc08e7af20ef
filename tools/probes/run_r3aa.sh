#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -s -m gpu -k "fp8_projections or l16" 2>&1 | grep -i "cosine\|passed\|failed\|error" 
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "colsum or cast_with or layernorm_backward or gradients_against" 2>&1 | tail -2
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$1', d['value'], d['ms_per_step'], r.get('frac'), d['dtype'][:40])"; }
python bench.py --model l16-fp8-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3 2>/dev/null | tail -1 | line c5_fp8
python bench.py --model l16-fp8-msclips --precision fp8-qkv --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3 2>/dev/null | tail -1 | line c5_fp8qkv
python bench.py --model l16-fp8-msclips --precision bf16 --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3 2>/dev/null | tail -1 | line c5_bf16
for bn in frozen batch; do python bench.py --train --bn $bn --no-cpu-baseline --no-pmc --no-probe --steps 15 --warmup 5 2>/dev/null | tail -1 | line train_$bn; done
