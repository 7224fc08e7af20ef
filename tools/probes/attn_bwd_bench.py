import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from msclip_amd import hip
for (B, L, causal) in [(512, 50, False), (512, 77, True), (256, 197, False)]:
    qkv = torch.randn(B * L, 2304, device="cuda").to(torch.bfloat16)
    o = torch.empty(B * L, 768, dtype=torch.bfloat16, device="cuda")
    hip.attention(qkv, o, B, L, 12, causal)
    do = torch.randn(B * L, 768, device="cuda").to(torch.bfloat16)
    dqkv = torch.zeros_like(qkv)
    for _ in range(3): hip.attention_bwd(qkv, o, do, dqkv, B, L, 12, causal)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): hip.attention_bwd(qkv, o, do, dqkv, B, L, 12, causal)
    e.record(); torch.cuda.synchronize()
    print(f"attention_bwd B={B} L={L} causal={causal}: {s.elapsed_time(e) / 20 * 1e3:8.1f} us")
