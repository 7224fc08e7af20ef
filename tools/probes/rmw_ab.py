"""In-place residual GEMMs (out_proj / c_proj shapes): every element against fp32 torch + timing.  MSCLIP_HIP_LIB picks the build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip

for name, M, N, K in (("out", 65024, 768, 768), ("proj", 65024, 768, 3072), ("out_ragged", 1000, 768, 768)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    r0 = torch.randn(M, N, generator=g).cuda()
    outs = []
    for rep in range(2):
        out = r0.clone()
        hip.gemm(x, w, out, bias=b, resid=out, resid_kind=hip.RESID_F32, tile=4)
        outs.append(out)
    assert torch.equal(outs[0], outs[1]), "not repeatable"
    worst = 0.0
    for a in range(0, M, 8192):
        ref = x[a:a + 8192].float() @ w.float().t() + b + r0[a:a + 8192]
        worst = max(worst, (outs[0][a:a + 8192] - ref).abs().max().item())
    assert worst < 2e-2, worst
    out = r0.clone()
    for _ in range(3):
        hip.gemm(x, w, out, bias=b, resid=out, resid_kind=hip.RESID_F32, tile=4)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        hip.gemm(x, w, out, bias=b, resid=out, resid_kind=hip.RESID_F32, tile=4)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"{os.path.basename(os.environ.get('MSCLIP_HIP_LIB') or 'shipped'):24s} {name:10s} max err {worst:.2e}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF")
