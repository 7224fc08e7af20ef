"""msclip_layernorm_bwd at the packed C2 row counts (GPU box only): us per launch and TB/s of x, dy, dx."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip
C = 768
for M in (25600, 17664):
    x, dy, dx, g = torch.randn(M, C, device="cuda"), torch.randn(M, C, device="cuda"), torch.randn(M, C, device="cuda"), torch.randn(C, device="cuda")
    f = lambda: hip.layernorm_bwd(x, dy, g, dx, M)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"layernorm_bwd M={M}: {us:6.1f} us per call (kernel + parameter-gradient fold), {M * C * 16 / us / 1e6:5.2f} TB/s of x, dy read + dx read / written")
