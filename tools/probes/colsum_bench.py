import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from msclip_amd import hip
for (M, C, dt) in [(65024, 768, torch.float32), (25600, 768, torch.float32), (39424, 768, torch.float32), (1024, 1536, torch.float32),
                   (65024, 3072, torch.bfloat16), (512 * 112 * 112, 48, torch.bfloat16), (512 * 56 * 56, 192, torch.bfloat16),
                   (512 * 28 * 28, 384, torch.bfloat16)]:
    x = torch.randn(M, C, device="cuda").to(dt)
    for _ in range(3): y = hip.colsum(x)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): y = hip.colsum(x)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    print(f"colsum {M}x{C} {str(dt)[6:]}: {us:8.1f} us  {M * C * x.element_size() / us / 1e6:6.2f} TB/s")
