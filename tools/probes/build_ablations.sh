#!/bin/bash
# Build probe copies of libmsclip_hip.so with -D knobs on ONE source file (A/B runs in one GPU call; never shipped).
#   usage: [ABL_FILE=front] build_ablations.sh name1 "-DX -DY" name2 "-DZ" ...     (default file: gemm)
set -u
SRC=/root/repo/msclip_amd/csrc   # -D switches act on the PRODUCT source of the chosen file (round 4: the instrumented snapshots of rounds 1-3 are gone)
OUT=/root/repo/tools/probes
FILE=${ABL_FILE:-gemm}
PSRC=$SRC/$FILE.hip
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-value"
OTHERS=$(ls $SRC/build/*.o | grep -v "/$FILE.o")
build() {  # name, defines
  /opt/rocm/bin/hipcc $F $2 -I$SRC -c $PSRC -o $OUT/abl_$1.o 2>&1 | grep -E "error" | head -3
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libgemm_$1.so $OUT/abl_$1.o $OTHERS || echo "BUILD FAILED $1"
  rm -f $OUT/abl_$1.o
}
rm -f $OUT/libgemm_*.so
while [ $# -ge 2 ]; do build "$1" "$2" & shift 2; done
wait
ls $OUT/*.so
