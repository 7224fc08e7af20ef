R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "fused_qkv" 2>&1 | tail -15
for i in 1 2; do for f in 0 1; do
MSCLIP_FUSED_QKV_ATTN=$f python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('fused=$f', r['ms_per_step'], r['value'], r['loss'])"
done; done | tee $O/r5h_fused_step_ab.txt
