"""Bitwise comparison of the GEMM main-loop variants on one problem (probe): python tools/probes/cmp_tiles.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip
g = torch.Generator().manual_seed(0)
M, N, K = 4096, 768, 768
x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).cuda()
b = torch.randn(N, generator=g).cuda()
r32 = torch.randn(M, N, generator=g).cuda()
r16 = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
for name, kw, dt in [("bias", dict(bias=b), torch.bfloat16), ("gelu", dict(bias=b, act=hip.ACT_QUICKGELU), torch.bfloat16),
                     ("resid32", dict(bias=b, resid_kind=hip.RESID_F32), torch.float32),
                     ("resid16relu", dict(bias=b, resid=r16, resid_kind=hip.RESID_BF16, act=hip.ACT_RELU), torch.bfloat16),
                     ("alpha", dict(bias=b, alpha=0.37), torch.float32)]:
    outs = {}
    for t in (1, 2, 4):
        out = r32.clone() if name == "resid32" else torch.zeros(M, N, dtype=dt, device="cuda")
        k = dict(kw)
        if name == "resid32":
            k["resid"] = out
        hip.gemm(x, w, out, tile=t, **k)
        outs[t] = out.float().clone()
    d12 = (outs[1] - outs[2]).abs().max().item()
    d14 = (outs[1] - outs[4]).abs()
    print(f"{name:12s} max|t1-t2| = {d12:.3e}   max|t1-t4| = {d14.max().item():.3e}  mismatching elements {int((d14 > 0).sum())}"
          f"  first bad {torch.nonzero(d14 > 0)[:3].tolist()}")
