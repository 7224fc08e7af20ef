R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -3
python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or conv" 2>&1 | tail -2
tr() { python bench.py --train --bn $1 --no-cpu-baseline --no-pmc --no-probe --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for i in 1 2 3; do
  echo -n "frozen relu pass  "; MSCLIP_RELU_BWD_PASS=1 tr frozen
  echo -n "frozen relu fused "; tr frozen
done
for i in 1 2; do
  echo -n "batch relu pass  "; MSCLIP_RELU_BWD_PASS=1 tr batch
  echo -n "batch relu fused "; tr batch
done
