"""Reference point only (not used by the product): vendor GEMM (hipBLASLt via torch) on the projection shapes."""
import torch
for (M, N, K) in [(65024, 2304, 768), (65024, 768, 768), (65024, 3072, 768), (65024, 768, 3072), (8192, 8192, 8192), (4096, 4096, 4096)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    for _ in range(3): y = torch.nn.functional.linear(x, w)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): y = torch.nn.functional.linear(x, w)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"torch.linear bf16 M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF (no bias/activation/residual epilogue)")
