R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for sz in 0 4194304 16777216; do
  if [ $sz = 0 ]; then unset HSA_KERNARG_POOL_SIZE; else export HSA_KERNARG_POOL_SIZE=$sz; fi
  echo "== HSA_KERNARG_POOL_SIZE=$sz"
  python tools/probes/queue_depth_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-260
  python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --no-probe --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('train frozen', r['ms_per_step'])"
  python bench.py --train --bn batch --no-cpu-baseline --no-pmc --no-probe --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('train batch', r['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-pmc --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('C2 forward', r['ms_per_step'])"
done
