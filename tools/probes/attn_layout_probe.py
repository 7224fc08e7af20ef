"""Does the attention kernel run faster when a (sample, head)'s q | k | v are sequential in memory?  Same kernel, same
bytes and FLOPs: (a) the model's layout (token rows of 3*H*64 values, one head = 128-byte pieces at a 4.6 KB stride),
(b) heads = 1 with nsamples = B*H (token rows of 192 values: every pair reads one sequential block)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from msclip_amd import hip
B, H = 512, 12
for L, causal in ((77, True), (50, False)):
    a = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
    oa = torch.empty(B * L, H * 64, dtype=torch.bfloat16, device="cuda")
    b = torch.randn(B * H * L, 192, device="cuda").to(torch.bfloat16)
    ob = torch.empty(B * H * L, 64, dtype=torch.bfloat16, device="cuda")
    for name, fn in (("model layout", lambda: hip.attention(a, oa, B, L, H, causal)),
                     ("sequential  ", lambda: hip.attention(b, ob, B * H, L, 1, causal))):
        for _ in range(5): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 50 * 1e3
        mb = B * L * H * 64 * 2 * 4 / 1e6
        print(f"L={L} causal={causal} {name}: {us:7.1f} us  {mb / us:6.2f} TB/s".replace("TB/s", "TB/s (MB/us)"))
