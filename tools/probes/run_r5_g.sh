R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_qkv" 2>&1 | tail -15
timeout 300 python tools/probes/fused_bench.py 2>&1 | tee $O/r5g_fused_bench.txt | tail -16
