R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
for lib in product occ4 product occ4; do
  if [ $lib = product ]; then unset MSCLIP_HIP_LIB; else export MSCLIP_HIP_LIB=$R/tools/probes/libgemm_$lib.so; fi
  echo "== $lib"; python tools/attn_bench.py 2>/dev/null | head -2
done | tee $O/r5l_attn_occ4.txt
for i in 1 2 3; do for lib in product occ4; do
  if [ $lib = product ]; then unset MSCLIP_HIP_LIB; else export MSCLIP_HIP_LIB=$R/tools/probes/libgemm_$lib.so; fi
  python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', r['ms_per_step'], r['value'])"
done; done | tee -a $O/r5l_attn_occ4.txt
