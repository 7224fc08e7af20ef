# usage: run_bench_ab.sh variant...   (bench.py with probe builds of the library, alternating, same box)
R=$GRAFT_REPO_ROOT; cd $R
for pass in 1 2; do
for v in "$@"; do
  if [ "$v" = "shipped" ]; then L=""; else L=$R/tools/probes/libgemm_$v.so; fi
  MSCLIP_HIP_LIB=$L python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['avg_launch_us'])"
done
done
