R=$GRAFT_REPO_ROOT; cd $R
MSCLIP_HIP_LIB=$R/tools/probes/libgemm_noepi.so timeout 300 python bench.py --no-cpu-baseline --no-pmc --prefill-random --shapes > gpurun_out/r2b_noepi.json 2> gpurun_out/r2b_noepi.err
tail -5 gpurun_out/r2b_noepi.err
python - <<EOF
import json
r=json.loads(open("gpurun_out/r2b_noepi.json").read().strip().splitlines()[-1]); ro=r["roofline"]
print("noepi", r["value"], r["ms_per_step"], ro["achieved"], ro["avg_launch_us"])
for s in ro["shapes"][:4]: print("   ", s["M"], s["N"], s["K"], s["avg_us"], s["tflops"])
EOF
