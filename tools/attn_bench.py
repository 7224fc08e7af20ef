"""Attention kernel micro-benchmark at the bench shapes (GPU box only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msclip_amd import hip
for (B, L, causal) in [(512, 50, False), (512, 77, True), (256, 197, False)]:
    qkv = torch.randn(B * L, 2304, device="cuda").to(torch.bfloat16)
    out = torch.empty(B * L, 768, dtype=torch.bfloat16, device="cuda")
    for _ in range(3): hip.attention(qkv, out, B, L, 12, causal)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): hip.attention(qkv, out, B, L, 12, causal)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    mb = (qkv.numel() + out.numel()) * 2 / 1e6
    print(f"B={B} L={L} causal={causal}: {us:8.1f} us   {mb/us*1e6/1e6:5.2f} TB/s of q,k,v,o traffic")
