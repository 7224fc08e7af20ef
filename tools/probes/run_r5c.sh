R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -3

tr() { python bench.py --train --bn $1 --no-cpu-baseline --no-pmc --no-probe --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for i in 1 2 3; do echo -n "batch "; tr batch; done
echo -n "frozen "; tr frozen
