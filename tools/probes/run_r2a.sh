# round-2 first GPU pass: new tests, bench lines for C2 / C3 / C4-per-rank with live PMC, kernel stats
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2a_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2a_pytest.log
timeout 900 python bench.py --shapes 2>gpurun_out/r2a_bench_b32.err | tail -1 > gpurun_out/r2a_bench_b32.json; cut -c1-400 gpurun_out/r2a_bench_b32.json
timeout 600 python bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --shapes 2>gpurun_out/r2a_bench_b16.err | tail -1 > gpurun_out/r2a_bench_b16.json; cut -c1-300 gpurun_out/r2a_bench_b16.json
timeout 600 python bench.py --batch 1024 --no-cpu-baseline --shapes 2>gpurun_out/r2a_bench_b1024.err | tail -1 > gpurun_out/r2a_bench_b1024.json; cut -c1-300 gpurun_out/r2a_bench_b1024.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2a_prof_b32 -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc > $R/gpurun_out/r2a_prof_b32.log 2>&1
cd $R; python tools/gemm_bench.py --tiles 4 > gpurun_out/r2a_gemm_bench.log 2>&1; tail -12 gpurun_out/r2a_gemm_bench.log
