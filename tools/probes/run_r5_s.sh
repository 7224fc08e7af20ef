R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "attention" 2>&1 | tail -3
python tools/probes/attn_bwd_bench.py 2>&1 | tail -3 | tee gpurun_out/r5s_attn_bwd.txt
for i in 1 2; do python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"; done
