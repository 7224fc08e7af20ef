cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python -u $R/tools/gemm_bench.py --tiles 4 --full 2>&1 | grep -v amdgpu > $R/gpurun_out/epi.log
cd $R; python -m pytest tests/test_gpu_kernels.py -x -q -k gemm 2>&1 | tail -3 >> $R/gpurun_out/epi.log
bash $R/tools/probes/run_variants.sh "--tiles 4" base rows16 >> $R/gpurun_out/epi.log 2>&1
bash $R/tools/probes/run_bench_ab.sh base rows16 >> $R/gpurun_out/epi.log 2>&1
cd /tmp; for s in qkv fc; do MSCLIP_HIP_LIB=$R/tools/probes/libgemm_trace.so python $R/tools/probes/pp_trace.py $s; done > $R/gpurun_out/trace.log 2>&1
