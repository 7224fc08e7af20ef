#!/bin/bash
# Build probe copies of libmsclip_hip.so with -D knobs (A/B runs of the GEMM main loop in one GPU call; never shipped).
#   usage: build_ablations.sh name1 "-DX -DY" name2 "-DZ" ...
set -u
SRC=/root/repo/msclip_amd/csrc
OUT=/root/repo/tools/probes
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-value"
build() {  # name, defines
  /opt/rocm/bin/hipcc $F $2 -c $SRC/gemm.hip -o $OUT/gemm_$1.o 2>&1 | grep -E "error" | head -3
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libgemm_$1.so $OUT/gemm_$1.o $SRC/build/gemm_small.o $SRC/build/api.o $SRC/build/attention.o $SRC/build/rows.o $SRC/build/conv.o $SRC/build/loss.o || echo "BUILD FAILED $1"
  rm -f $OUT/gemm_$1.o
}
rm -f $OUT/libgemm_*.so
while [ $# -ge 2 ]; do build "$1" "$2" & shift 2; done
wait
ls $OUT/*.so
