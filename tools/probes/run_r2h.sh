R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/r2h_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2h_pytest.log | cut -c1-200
timeout 600 python bench.py --train-slice --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-probe 2>gpurun_out/r2h_bench_train.err | tail -1 > gpurun_out/r2h_bench_train.json; cut -c1-260 gpurun_out/r2h_bench_train.json; tail -2 gpurun_out/r2h_bench_train.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2h_prof_train -o run -- python $R/bench.py --train-slice --steps 3 --warmup 1 --no-cpu-baseline --no-probe --no-pmc > $R/gpurun_out/r2h_prof_train.log 2>&1
head -12 $R/gpurun_out/r2h_prof_train/run_kernel_stats.csv | cut -c1-150
