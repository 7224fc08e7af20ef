# same-box alternating A/B of library builds on the default bench (short lines)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "" tools/probes/libgemm_fix.so tools/probes/libgemm_nont.so; do
  MSCLIP_HIP_LIB=$lib python bench.py --steps 30 --warmup 5 --no-pmc --no-cpu-baseline --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('${lib:-product}', r['ms_per_step'], r['roofline']['frac'], r['roofline']['isolated']['frac'])"
done; done
