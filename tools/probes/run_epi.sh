cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python -u $R/tools/gemm_bench.py --tiles 4 --full 2>&1 | grep -v amdgpu | grep "full check\|rror" > $R/gpurun_out/epi.log
cd $R; python -m pytest tests/test_gpu_kernels.py -x -q -k gemm 2>&1 | tail -3 >> $R/gpurun_out/epi.log
bash $R/tools/probes/run_variants.sh "--tiles 4 --square" head m16 2>&1 | grep "==\|^sq8" >> $R/gpurun_out/epi.log
bash $R/tools/probes/run_variants.sh "--tiles 4" head m16 2>&1 | grep "==\|^qkv \|^fc \|^out \|^proj " >> $R/gpurun_out/epi.log
bash $R/tools/probes/run_bench_ab.sh head m16 >> $R/gpurun_out/epi.log 2>&1
