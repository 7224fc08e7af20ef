#!/bin/bash
# round 4: in-place residual epilogue as L2 atomics (-DPP_ATOMIC_RMW) against the shipped load / add / store form
R=$GRAFT_REPO_ROOT; cd $R
for pass in 1 2; do
  for v in shipped atomic; do
    if [ "$v" = "shipped" ]; then L=""; else L=$R/tools/probes/libgemm_$v.so; fi
    MSCLIP_HIP_LIB=$L python tools/probes/rmw_ab.py 2>&1 | grep -v amdgpu
  done
done
bash tools/probes/run_bench_ab.sh shipped atomic
