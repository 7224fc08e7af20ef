R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
python tools/probes/fused_calib.py > $O/r5b_fused_calib.txt 2>&1; tail -12 $O/r5b_fused_calib.txt
python tools/probes/f8_epi_bench.py 2>&1 | tail -1 > $O/r5b_f8_epi.jsonl
MSCLIP_HIP_LIB=$R/tools/probes/libgemm_noepi.so python tools/probes/f8_epi_bench.py 2>&1 | tail -1 >> $O/r5b_f8_epi.jsonl
cat $O/r5b_f8_epi.jsonl
: > $O/r5b_ab.txt
for i in 1 2; do
  for lib in product qkvtemporal; do
    if [ $lib = product ]; then unset MSCLIP_HIP_LIB; else export MSCLIP_HIP_LIB=$R/tools/probes/libgemm_$lib.so; fi
    python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', r['ms_per_step'], r['value'])" >> $O/r5b_ab.txt
  done
done
unset MSCLIP_HIP_LIB
cat $O/r5b_ab.txt
