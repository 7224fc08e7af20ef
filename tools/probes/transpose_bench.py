import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from msclip_amd import hip
for (M, C) in [(65024, 3072), (65024, 768), (65024, 2304), (25600 * 16, 192)]:
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    Mp = (M + 127) // 128 * 128
    for _ in range(3): y = hip.transpose_bf16(x, M, Mp)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): y = hip.transpose_bf16(x, M, Mp)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    print(f"transpose {M}x{C}: {us:8.1f} us  {2 * M * C * 2 / us / 1e6:6.2f} TB/s (read + write)")
    assert torch.equal(y[:, :M], x.t())
