cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python $R/tools/gemm_bench.py --tiles 4 --full > $R/gpurun_out/epi.log 2>&1
python $R/tools/gemm_bench.py --tiles 2 4 >> $R/gpurun_out/epi.log 2>&1
for s in qkv fc out; do MSCLIP_HIP_LIB=$R/tools/probes/libgemm_trace.so python $R/tools/probes/pp_trace.py $s; done > $R/gpurun_out/trace.log 2>&1
cd $R; python -m pytest tests/test_gpu_kernels.py -x -q -k gemm 2>&1 | tail -3 >> $R/gpurun_out/epi.log
