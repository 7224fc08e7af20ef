# round 3: gemm_pp2 first contact: correctness, micro-benchmark against the ping-pong kernel, in-model A/B, delay modes
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > gpurun_out/r3b_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3b_tests.log
tail -12 gpurun_out/r3b_tests.log
for dm in 1 0 2; do
  echo "== gemm_bench tiles 4 8, MSCLIP_PP2_DELAY=$dm"
  MSCLIP_PP2_DELAY=$dm timeout 600 python tools/gemm_bench.py --tiles 4 8 2>&1 | tail -11
done
for v in "MSCLIP_GEMM_PP2=0" "MSCLIP_GEMM_PP2=1" "MSCLIP_GEMM_PP2=1 MSCLIP_PP2_DELAY=0" "MSCLIP_GEMM_PP2=1 MSCLIP_PP2_DELAY=2" "MSCLIP_GEMM_PP2=1 MSCLIP_PP2_DELAY_UNIT=8" "MSCLIP_GEMM_PP2=1 MSCLIP_PP2_DELAY_UNIT=32"; do
  echo "== bench $v"; env $v python bench.py --no-cpu-baseline --no-pmc --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r.get('roofline',{}).get('frac'), r.get('roofline',{}).get('avg_launch_us'))"
done
