R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
for lib in product ntw ntx ntwx; do
  if [ $lib = product ]; then unset MSCLIP_HIP_LIB; else export MSCLIP_HIP_LIB=$R/tools/probes/libgemm_$lib.so; fi
  FOLD_M=43264 python tools/probes/fold_bench.py 2>/dev/null | grep JSON | python -c "
import json,sys
r=json.loads(sys.stdin.read()[5:]); u=r['us']; print('$lib', {k:u[k] for k in ('qkv fold 1seg','fc fold 1seg','out producer','proj producer')}, 'sum', round(sum(u[k] for k in ('qkv fold 1seg','fc fold 1seg','out producer','proj producer')),1))"
done | tee $O/r5j_nt_dma.txt
for i in 1 2; do for lib in product ntw ntx; do
  if [ $lib = product ]; then unset MSCLIP_HIP_LIB; else export MSCLIP_HIP_LIB=$R/tools/probes/libgemm_$lib.so; fi
  python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', r['ms_per_step'], r['value'])"
done; done | tee -a $O/r5j_nt_dma.txt
