#!/bin/bash
# Build libmsclip_hip.so in-tree for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wall -Wno-unused-function"
mkdir -p build
objs=()
pids=()
for f in gemm gemm_small attention qkv_attention attention_bwd backward backward_conv rows conv front loss pack api plan comm; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm_epilogue.h -nt build/$f.o ] || [ plan.h -nt build/$f.o ] || [ ../../include/msclip_hip.h -nt build/$f.o ]; then
    # pack.hip restates tensor algebra that must come out bitwise (IEEE division / square root, no contraction): no fast-math there
    if [ $f = pack ]; then FL="${FLAGS/-ffast-math -fno-finite-math-only/-fno-fast-math -ffp-contract=off}"; else FL="$FLAGS"; fi
    $HIPCC $FL -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
  objs+=(build/$f.o)
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmsclip_hip.so "${objs[@]}"
echo "built $(pwd)/libmsclip_hip.so"
