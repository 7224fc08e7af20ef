#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "last_block or golden or graph" 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --no-pmc --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['ms_per_step'])"; done
python bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'])"
cd /tmp; MSCLIP_CONV_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lq -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-probe --no-pmc --steps 10 > /dev/null 2>&1; grep -i "lastq\|attn_kernel" /tmp/lq/run_kernel_stats.csv | cut -c1-160
