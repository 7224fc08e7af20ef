R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in w4drop w4raw; do
MSCLIP_HIP_LIB=$R/tools/probes/libgemm_$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2e_prof_$v -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc --prefill-random > $R/gpurun_out/r2e_prof_$v.log 2>&1
echo $v; grep "gemm_w4" $R/gpurun_out/r2e_prof_$v/run_kernel_stats.csv | cut -c1-120
done
