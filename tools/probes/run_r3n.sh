#!/bin/bash
# dynamic tile lists of the ping-pong GEMM: tests, then A/B against MSCLIP_GEMM_DYN=0
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "golden or last_block" 2>&1 | tail -3
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; }
for i in 1 2; do
  MSCLIP_GEMM_DYN=0 python bench.py --no-cpu-baseline --no-pmc --steps 40 --warmup 10 2>/dev/null | tail -1 | line static
  python bench.py --no-cpu-baseline --no-pmc --steps 40 --warmup 10 2>/dev/null | tail -1 | line dyn
done
MSCLIP_GEMM_DYN=0 python bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 30 --warmup 10 2>/dev/null | tail -1 | line b16_static
python bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 30 --warmup 10 2>/dev/null | tail -1 | line b16_dyn
MSCLIP_GEMM_DYN=0 python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --steps 12 --warmup 5 2>/dev/null | tail -1 | line train_static
python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --steps 12 --warmup 5 2>/dev/null | tail -1 | line train_dyn
