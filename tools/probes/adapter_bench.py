"""Adapter pass (msclip_adapter_combine_ln_stats) at the C2 / C3 shapes: workgroup-per-sample form vs wave-per-grid-row form (GPU box only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip
BF = torch.bfloat16
for B, g, C in ((512, 7, 768), (256, 14, 768), (1024, 7, 768)):
    L = g * g + 1
    x, t = torch.randn(B * L, C, device="cuda"), torch.randn(B * g * g, C, device="cuda")
    dww, dwb = torch.randn(9, C, device="cuda") * 0.3, torch.randn(C, device="cuda") * 0.1
    ga, be, g1, b1 = (torch.randn(C, device="cuda") * 0.1 + 1 for _ in range(4))
    xa, lno = torch.empty(B * L, C, device="cuda"), torch.empty(B * L, C, dtype=BF, device="cuda")
    cen, rs = torch.empty(B * L, device="cuda"), torch.empty(B * L, 2, device="cuda")
    mb = (B * L * C * (4 + 4 + 2) + B * g * g * C * 4) / 1e6
    for form in ("1", "0", "1", "0"):
        os.environ["MSCLIP_ADAPTER_SAMPLE"] = form
        f = lambda: hip.adapter_combine_ln_stats(x, t, dww, dwb, ga, be, xa, g1, b1, lno, cen, rs, B, L, g, True)
        for _ in range(3): f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        print(f"B={B} g={g} C={C} form={'sample' if form == '1' else 'gridrow'}: {us:7.1f} us  {mb / us:5.2f} TB/s of x, t in + stream, operand out")
