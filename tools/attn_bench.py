"""Attention kernel micro-benchmark at the bench shapes (GPU box only): us per launch and q|k|v|o bytes / time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msclip_amd import hip
for (B, L, H, causal) in [(512, 50, 12, False), (512, 77, 12, True), (256, 197, 12, False), (256, 257, 16, False), (256, 257, 16, True),
                          (256, 258, 16, False), (256, 260, 16, False), (256, 261, 16, False), (256, 256, 16, False)]:
    D = H * 64
    qkv = torch.randn(B * L, 3 * D, device="cuda").to(torch.bfloat16)
    out = torch.empty(B * L, D, dtype=torch.bfloat16, device="cuda")
    for _ in range(3): hip.attention(qkv, out, B, L, H, causal)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): hip.attention(qkv, out, B, L, H, causal)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    mb = (qkv.numel() + out.numel()) * 2 / 1e6
    print(f"B={B} L={L} H={H} causal={causal}: {us:8.1f} us   {mb/us*1e6/1e6:5.2f} TB/s of q,k,v,o traffic")
