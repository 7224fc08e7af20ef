R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
python tools/probes/host_profile.py --train --bn frozen --steps 6 --top 25 2>&1 | tee $O/r5q_host_profile_train_frozen.txt | head -32
for i in 1 2; do for bn in frozen batch; do python bench.py --train --bn $bn --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$bn', r['ms_per_step'], r['value'])"; done; done
python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('C2', r['ms_per_step'], r['value'])"
