"""Train-mode BatchNorm passes at the conv side's map shapes (GPU box only): us per call and TB/s of the bytes each pass moves.
    python tools/probes/bn_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip

BF = torch.bfloat16


def t(f, n=10):
    for _ in range(2):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, C in ((6422528, 48), (1605632, 96), (401408, 192), (100352, 384), (25088, 768)):
    x = torch.randn(M, C, device="cuda")
    g, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    out = torch.empty(M, C, dtype=BF, device="cuda")
    res = torch.randn(M, C, device="cuda").to(BF)
    dy = torch.randn(M, C, device="cuda").to(BF)
    dx = torch.empty(M, C, dtype=BF, device="cuda")
    mean, var, rstd, scale, shift = hip.bn_stats(x, M, gamma=g, beta=b)
    n = M * C
    us = t(lambda: hip.bn_stats(x, M, gamma=g, beta=b))
    print(f"[{M:8d} x {C:3d}] bn_stats        {us:7.1f} us  {n * 4 / us / 1e6:5.2f} TB/s (x fp32 read)")
    us = t(lambda: hip.bn_apply(x, scale, shift, out, M, relu=True))
    print(f"[{M:8d} x {C:3d}] bn_apply        {us:7.1f} us  {n * 6 / us / 1e6:5.2f} TB/s (x read, bf16 written)")
    us = t(lambda: hip.bn_apply(x, scale, shift, out, M, relu=True, resid=res))
    print(f"[{M:8d} x {C:3d}] bn_apply+resid  {us:7.1f} us  {n * 8 / us / 1e6:5.2f} TB/s")
    us = t(lambda: hip.bn_bwd(dy, x, mean, rstd, g, dx, M))
    print(f"[{M:8d} x {C:3d}] bn_bwd (2 pass) {us:7.1f} us  {n * 14 / us / 1e6:5.2f} TB/s (dy + x read twice, dx written)")
    us = t(lambda: hip.relu_bwd(dy, out, dx))
    print(f"[{M:8d} x {C:3d}] relu_bwd        {us:7.1f} us  {n * 6 / us / 1e6:5.2f} TB/s")
