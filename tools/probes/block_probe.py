"""Which launches block the host while the GPU queue is deep?  (GPU box only.)
Enqueue ~60 ms of GEMMs, then time single host calls: a small-argument launch, msclip_adamw_multi (3.9 KB of kernel
arguments), torch.cat, torch.empty, an elementwise ATen op."""
import sys
import time

import torch

sys.path.insert(0, ".")
from msclip_amd import hip                                                 # noqa: E402

dev = torch.device("cuda", 0)
a = torch.randn(65024, 768, device=dev).to(torch.bfloat16)
w = torch.randn(3072, 768, device=dev).to(torch.bfloat16)
o = torch.empty(65024, 3072, dtype=torch.bfloat16, device=dev)
x = torch.randn(4096, 768, device=dev)
g, b = torch.ones(768, device=dev), torch.zeros(768, device=dev)
lo = torch.empty(4096, 768, dtype=torch.bfloat16, device=dev)
ps = [torch.randn(100000, device=dev) for _ in range(8)]
items = [(p, torch.randn_like(p), torch.zeros_like(p), torch.zeros_like(p), 1e-4, 0.1) for p in ps]
plan = hip.AdamwPlan(items)
c1, c2 = torch.randn(48, 3, 3, 3, device=dev), torch.randn(48, 3, 3, 3, device=dev)
s1 = torch.randn(48, device=dev)


def fill(n=200):
    for _ in range(n):
        hip.gemm(a, w, o)


def timed(name, fn, deep):
    torch.cuda.synchronize()
    if deep:
        fill()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"  {name:44s} queue {'deep ' if deep else 'empty'}: host {1e3 * (t1 - t0):8.3f} ms")


fill(20)
torch.cuda.synchronize()
t0 = time.perf_counter(); fill(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"200 GEMMs: host issue {1e3 * (t1 - t0):.1f} ms, GPU {1e3 * (t2 - t0):.1f} ms")
for deep in (False, True, True):
    timed("layernorm (small kernel arguments)", lambda: hip.layernorm(x, g, b, lo, 4096), deep)
    timed("msclip_adamw_multi (3.9 KB arguments)", lambda: plan.run(0.9, 0.999, 1e-8, 3), deep)
    timed("torch.cat of two [48,3,3,3] products", lambda: torch.cat([c1 * s1[:, None, None, None], c2 * s1[:, None, None, None]], 0), deep)
    timed("torch.cat of two plain tensors", lambda: torch.cat([c1, c2], 0), deep)
    timed("torch.empty + mul", lambda: c1 * 2.0, deep)
    timed("torch.outer", lambda: torch.outer(s1, s1), deep)
    timed("tensor.sum()", lambda: c1.sum(), deep)
    timed("torch.cuda.Event record", lambda: torch.cuda.Event().record(), deep)
