"""Host time per launch (no profiler): a few kernels of the library, each launched back to back without synchronising."""
import sys
import time

import torch

sys.path.insert(0, ".")
from msclip_amd import hip                                               # noqa: E402

dev = torch.device("cuda")
hip.use_compute_stream(dev) if len(sys.argv) > 1 and sys.argv[1] == "compute" else None
M, D = 8192, 768
x = torch.randn(M, D, device=dev).to(torch.bfloat16)
w = torch.randn(3 * D, D, device=dev).to(torch.bfloat16) * 0.02
b = torch.zeros(3 * D, device=dev)
out = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
xf = torch.randn(M, D, device=dev)
g = torch.ones(D, device=dev)
ln = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
small = torch.randn(256, 64, device=dev).to(torch.bfloat16)
wsmall = torch.randn(64, 64, device=dev).to(torch.bfloat16)
osmall = torch.empty(256, 64, dtype=torch.bfloat16, device=dev)


def bench(name, fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:44s} host {1e6 * (t1 - t0) / n:7.1f} us / call   (GPU drained after {1e3 * (t2 - t1):6.1f} ms)")


bench("gemm_pp (8192 x 2304 x 768)", lambda: hip.gemm(x, w, out, bias=b))
bench("gemm small (dense128 / stream)", lambda: hip.gemm(small, wsmall, osmall))
bench("layernorm", lambda: hip.layernorm(xf, g, b[:D], ln, M))
bench("colsum", lambda: hip.colsum(xf, M=M))
bench("torch add_", lambda: xf.add_(1.0))
bench("gemm_pp tiny M=256 tile=4", lambda: hip.gemm(x[:256], w, out[:256], bias=b, tile=4))
