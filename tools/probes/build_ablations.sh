#!/bin/bash
# Build ablated copies of libmsclip_hip.so (A/B probes for the GEMM main loop; never shipped).
set -u
SRC=/root/repo/msclip_amd/csrc
OUT=/root/repo/tools/probes
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-value"
build() {  # name, defines
  /opt/rocm/bin/hipcc $F $2 -shared -o $OUT/libgemm_$1.so $SRC/gemm.hip $SRC/api.hip $SRC/attention.hip $SRC/rows.hip $SRC/conv.hip $SRC/loss.hip 2>&1 | grep -E "error" | head -3; [ ${PIPESTATUS[0]} -eq 0 ] || echo "BUILD FAILED $1"
}
build noEPI "-DMSCLIP_ABLATE_EPI" &
build noDMA "-DMSCLIP_ABLATE_DMA" &
build noDS "-DMSCLIP_ABLATE_DSREAD" &
build noDMADS "-DMSCLIP_ABLATE_DMA -DMSCLIP_ABLATE_DSREAD" &
build noWAIT "-DMSCLIP_ABLATE_DMAWAIT" &
build noBAR "-DMSCLIP_ABLATE_BARRIER" &
build noDMADSEPI "-DMSCLIP_ABLATE_DMA -DMSCLIP_ABLATE_DSREAD -DMSCLIP_ABLATE_EPI" &
wait
ls -la $OUT/*.so
