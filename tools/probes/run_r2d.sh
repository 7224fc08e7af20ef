R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python tools/gemm_bench.py --tiles 4 7 > gpurun_out/r2d_gemm_bench.log 2>&1; tail -11 gpurun_out/r2d_gemm_bench.log
timeout 300 python bench.py --no-cpu-baseline --no-pmc --shapes 2>/dev/null | tail -1 > gpurun_out/r2d_bench.json
python - <<EOF
import json
r=json.load(open("gpurun_out/r2d_bench.json")); ro=r["roofline"]
print("step", r["value"], r["ms_per_step"], ro["achieved"], ro["avg_launch_us"], ro["launches_per_step"])
for s in ro["shapes"][:6]: print("   ", s["M"], s["N"], s["K"], s["avg_us"], s["tflops"])
EOF
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2d_prof -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc > $R/gpurun_out/r2d_prof.log 2>&1
head -8 $R/gpurun_out/r2d_prof/run_kernel_stats.csv | cut -c1-150
