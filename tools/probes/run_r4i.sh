# P1: ln_1 behind a lateral adapter applied inside the adapter kernel (image rows' stream stays in XA until out_proj); tests, then A/B
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -x -k "adapter or fold or layernorm_stats" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -3
ab() { python bench.py --no-cpu-baseline --no-pmc --no-hbm-kernels --no-probe "$@" 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"; }
for i in 1 2 3; do
  echo -n "pass  "; MSCLIP_ADAPTER_LN1_PASS=1 ab
  echo -n "fused "; ab
done
echo C3; for i in 1 2; do
  echo -n "pass  "; MSCLIP_ADAPTER_LN1_PASS=1 ab --model b16-yfcc-msclips --batch 256
  echo -n "fused "; ab --model b16-yfcc-msclips --batch 256
done
