cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python $R/tools/gemm_bench.py --tiles 4 --square --full > $R/gpurun_out/sq.log 2>&1
timeout 300 python $R/tools/gemm_bench.py --tiles 2 3 4 --square >> $R/gpurun_out/sq.log 2>&1
timeout 300 python $R/tools/gemm_bench.py --tiles 2 4 >> $R/gpurun_out/sq.log 2>&1
cat $R/gpurun_out/sq.log
