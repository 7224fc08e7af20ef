"""Timing of the adapter pooling kernel on the five stages of the conv branch (GPU box only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msclip_amd import hip
B = 512
for (C, k, H) in [(48, 16, 112), (96, 8, 56), (192, 4, 28), (384, 2, 14), (768, 1, 7)]:
    g = H // k
    top = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16)
    w = torch.randn(k * k, C, device="cuda")
    out = torch.empty(B * g * g, C, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        hip.dwpool(top, w, out, B, H, H, C, k)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        hip.dwpool(top, w, out, B, H, H, C, k)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    mb = top.numel() * 2 / 1e6
    print(f"dwpool C={C:4d} k={k:2d} {H}x{H}: {us:8.1f} us  {mb / us:6.2f} TB/s")

# lateral adapter combine (depthwise 3x3 over the token grid + t + LayerNorm)
for (Bq, g) in [(512, 7), (256, 14)]:
    L, C = g * g + 1, 768
    xin = torch.randn(Bq * L, C, device="cuda")
    t = torch.randn(Bq * g * g, C, device="cuda")
    dww, dwb = torch.randn(9, C, device="cuda") * 0.2, torch.randn(C, device="cuda") * 0.1
    gam, bet = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    xout = torch.empty_like(xin)
    for _ in range(3):
        hip.adapter_combine_ln(xin, t, dww, dwb, gam, bet, xout, Bq, L, g, True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        hip.adapter_combine_ln(xin, t, dww, dwb, gam, bet, xout, Bq, L, g, True)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    mb = (xin.numel() * 2 + t.numel()) * 4 / 1e6
    print(f"adapter B={Bq} g={g}: {us:8.1f} us  {mb / us:6.2f} TB/s of x, t, out traffic")
