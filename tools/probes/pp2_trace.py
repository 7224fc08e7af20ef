"""Cycle stamps of a P2_TRACE build of gemm_pp2_kernel (probe only): which workgroups share a CU, how their phases interleave.
    MSCLIP_HIP_LIB=tools/probes/libgemm_trace.so python tools/probes/pp2_trace.py <shape> [env knobs MSCLIP_PP2_*]
stamp ids: 1 phase A start, 2 after A's wait + barrier, 3 phase B start, 4 after B's wait + barrier, 5 epilogue start, 6 tile end."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip
from tools.gemm_bench import SHAPES, run
which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
N_ = 128
lib = ctypes.CDLL(hip.LIB_PATH)
for name, M, N, K, epi in SHAPES:
    if name.strip() != which:
        continue
    us, tf = run(name, M, N, K, epi, 8, iters=1)
    print(name, f"{us:.1f} us {tf:.1f} TF")
    nwg = 512
    buf = np.zeros(nwg * N_, dtype=np.uint64)
    assert lib.msclip_pp2_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    buf = buf.reshape(nwg, N_)
    hw = (buf[:, 0] & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (buf[:, 0] >> np.uint64(32)).astype(np.int64) & 0xf
    key = [(int(xcc[b]), int(hw[b] >> 8) & 0xff) for b in range(nwg)]
    slots = {}
    for b, k in enumerate(key):
        slots.setdefault(k, []).append(b)
    hist = {}
    for k, v in slots.items():
        hist[len(v)] = hist.get(len(v), 0) + 1
    print("workgroups per (xcc, se/sh/cu) key:", hist, " distinct keys:", len(slots))
    print("wave slot (HW_ID & 15) of wave 0 by block half:", np.bincount(hw[:256] & 15, minlength=4)[:4], np.bincount(hw[256:] & 15, minlength=4)[:4])
    ids = (buf >> np.uint64(56)).astype(int)
    t = (buf & np.uint64((1 << 56) - 1)).astype(np.int64)
    # per-workgroup summary over the stamps of the main loop: share of time between (1 -> 2) + (3 -> 4) = wait + barrier
    wait = comp = 0
    for b in range(nwg):
        n = int((ids[b, 1:] > 0).sum())
        for i in range(1, n):
            d = t[b, i + 1] - t[b, i] if i + 1 <= n else 0
            if ids[b, i] in (1, 3) and ids[b, i + 1] in (2, 4):
                wait += d
            elif ids[b, i] in (2, 4) and ids[b, i + 1] in (3, 1, 5):
                comp += d
    print(f"all workgroups: wait+barrier {wait / (wait + comp):.3f} of the main loop, rest (issue + reads + MFMAs) {comp / (wait + comp):.3f}")
    pairs = [v for v in slots.values() if len(v) == 2][:3]
    for v in pairs:
        t0 = min(t[v[0], 1], t[v[1], 1])
        for b in v:
            n = int((ids[b, 1:] > 0).sum())
            print(f"  block {b} hw {hw[b]:#x}: " + " ".join(f"{ids[b, i]}@{t[b, i] - t0}" for i in range(1, min(n + 1, 60))))
