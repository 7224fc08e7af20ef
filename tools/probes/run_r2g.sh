# round-2 measurement pass: full GPU suite, bench lines (C2 with live PMC + CPU baseline, C3, C4-per-rank, train slice),
# kernel stats + per-kernel HBM counters for C2 / C3 / C4-per-rank, PMC sets on the shipped ping-pong kernel
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2g_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2g_pytest.log
timeout 900 python bench.py --shapes 2>gpurun_out/r2g_bench_b32.err | tail -1 > gpurun_out/r2g_bench_b32.json; cut -c1-200 gpurun_out/r2g_bench_b32.json
timeout 600 python bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --shapes 2>gpurun_out/r2g_bench_b16.err | tail -1 > gpurun_out/r2g_bench_b16.json; cut -c1-200 gpurun_out/r2g_bench_b16.json
timeout 600 python bench.py --batch 1024 --no-cpu-baseline --shapes 2>gpurun_out/r2g_bench_b1024.err | tail -1 > gpurun_out/r2g_bench_b1024.json; cut -c1-200 gpurun_out/r2g_bench_b1024.json
timeout 600 python bench.py --train-slice --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-probe 2>gpurun_out/r2g_bench_train.err | tail -1 > gpurun_out/r2g_bench_train.json; cut -c1-300 gpurun_out/r2g_bench_train.json; tail -3 gpurun_out/r2g_bench_train.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2g_prof_b32 -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc > $R/gpurun_out/r2g_prof_b32.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2g_prof_b16 -o run -- python $R/bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-probe --no-pmc > $R/gpurun_out/r2g_prof_b16.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2g_prof_b1024 -o run -- python $R/bench.py --batch 1024 --no-cpu-baseline --no-probe --no-pmc > $R/gpurun_out/r2g_prof_b1024.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2g_prof_train -o run -- python $R/bench.py --train-slice --steps 3 --warmup 1 --no-cpu-baseline --no-probe --no-pmc > $R/gpurun_out/r2g_prof_train.log 2>&1
cd $R
bash tools/pmc_bench.sh r02b32
bash tools/pmc_bench.sh r02b16 --model b16-yfcc-msclips --batch 256
bash tools/pmc_gemm.sh 4 qkv r02pp_qkv > /dev/null 2>&1
bash tools/pmc_gemm.sh 4 proj r02pp_proj > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r02pp_qkv gemm_pp > gpurun_out/r2g_pmc_pp_qkv.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r02pp_proj gemm_pp > gpurun_out/r2g_pmc_pp_proj.txt 2>&1
head -3 gpurun_out/r2g_pmc_pp_qkv.txt
