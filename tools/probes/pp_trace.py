"""Print the cycle stamps a PP_TRACE build of the ping-pong GEMM leaves behind (probe only).
    MSCLIP_HIP_LIB=tools/probes/libgemm_trace.so python tools/probes/pp_trace.py <shape>"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip
from tools.gemm_bench import SHAPES, SQUARE, run
which = sys.argv[1] if len(sys.argv) > 1 else "out"
lib = ctypes.CDLL(hip.LIB_PATH)
for name, M, N, K, epi in SHAPES + SQUARE:
    if name.strip() != which:
        continue
    us, tf = run(name, M, N, K, epi, 4, iters=2)
    print(name, f"{us:.1f} us {tf:.1f} TF")
    buf = np.zeros(1024, dtype=np.uint64)
    assert lib.msclip_pp_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    for g in range(2):
        b = buf[g * 512:(g + 1) * 512]
        ids = (b >> np.uint64(56)).astype(int)
        t = (b & np.uint64((1 << 56) - 1)).astype(np.int64)
        n = int((ids > 0).sum())
        print(f"group {g}: {n} stamps; id:delta_cycles(100MHz ticks?)")
        line = []
        for i in range(n):
            line.append(f"{ids[i]}:{(t[i] - t[i-1]) if i else 0}")
        print(" ".join(line))
