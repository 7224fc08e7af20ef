R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -x -k "adapter or fold or layernorm_stats" 2>&1 | tail -3
python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -3
python tools/probes/host_profile.py frozen 2>&1 | grep -v "^$" | cut -c1-200 > gpurun_out/r4k_host_profile.txt
head -3 gpurun_out/r4k_host_profile.txt
for i in 1 2; do python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --no-probe --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('train frozen', r['ms_per_step'])"; done
python bench.py --train --bn batch --no-cpu-baseline --no-pmc --no-probe --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('train batch', r['ms_per_step'])"
