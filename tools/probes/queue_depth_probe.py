"""How far ahead of the GPU can the host get?  Issue N long kernels without synchronising and find the launch at which the host
starts to block (its per-launch time jumps from ~10 us to the kernel's duration)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from msclip_amd import hip                                                 # noqa: E402

dev = torch.device("cuda", 0)
a = torch.randn(65024, 768, device=dev).to(torch.bfloat16)
w = torch.randn(3072, 768, device=dev).to(torch.bfloat16)
o = torch.empty(65024, 3072, dtype=torch.bfloat16, device=dev)
x = torch.randn(65024, 768, device=dev)
g, b = torch.ones(768, device=dev), torch.zeros(768, device=dev)
lo = torch.empty(65024, 768, dtype=torch.bfloat16, device=dev)
ps = [torch.randn(4000000, device=dev) for _ in range(36)]
plan = hip.AdamwPlan([(p, torch.randn_like(p), torch.zeros_like(p), torch.zeros_like(p), 1e-4, 0.1) for p in ps])
y = torch.randn(8 << 20, device=dev)


def probe(name, fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for _ in range(n):
        fn()
        ts.append(time.perf_counter())
    t_issue = ts[-1] - ts[0]
    torch.cuda.synchronize()
    t_all = time.perf_counter() - ts[0]
    d = [b - a for a, b in zip(ts, ts[1:])]
    fast = sorted(d)[len(d) // 10]
    first_slow = next((i for i, v in enumerate(d) if v > 20 * fast and v > 100e-6), None)
    slow = [i for i, v in enumerate(d) if v > 20 * fast and v > 100e-6]
    print(f"{name}: {n} calls, host issue {1e3 * t_issue:.1f} ms, GPU done after {1e3 * t_all:.1f} ms; typical call {1e6 * fast:.1f} us; "
          f"first blocking call #{first_slow}, {len(slow)} blocking calls, first few at {slow[:8]}")


probe("gemm (230 B of kernel arguments, ~280 us)", lambda: hip.gemm(a, w, o), 6000)
probe("layernorm (small arguments, ~45 us)", lambda: hip.layernorm(x, g, b, lo, 65024), 12000)
probe("adamw_multi (3.9 KB arguments x 13 launches per call)", lambda: plan.run(0.9, 0.999, 1e-8, 3), 400)
probe("aten mul_ (in place, 32 MB)", lambda: y.mul_(1.0001), 12000)
