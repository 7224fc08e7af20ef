R=$GRAFT_REPO_ROOT; cd $R
python bench.py 2>/dev/null | tail -1 > $R/gpurun_out/bench_full.json
bash tools/pmc_bench.sh r01pp
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01pp -o run -- python $R/bench.py --no-cpu-baseline --no-probe > $R/gpurun_out/prof_r01pp.log 2>&1
cd $R; python tools/pmc_traffic.py gpurun_out/pmcb_r01pp "gemm_pp_kernel<0>" > gpurun_out/traffic_pp.json; head -c 600 gpurun_out/traffic_pp.json; cat gpurun_out/bench_full.json
