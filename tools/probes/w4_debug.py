"""Error map of the 4-wave GEMM per 16 x 16 block (GPU box only): python tools/probes/w4_debug.py M N K"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 256, 576)
ACT = int(sys.argv[4]) if len(sys.argv) > 4 else 0
BIAS = int(sys.argv[5]) if len(sys.argv) > 5 else 0
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
b = torch.randn(N, generator=g).cuda() if BIAS else None
hip.gemm(x, w, out, tile=7, act=ACT, bias=b)
ref = x.float() @ w.float().t() + (b if BIAS else 0)
if ACT == 1:
    ref = ref * torch.sigmoid(1.702 * ref)
if ACT == 2:
    ref = torch.relu(ref)
print("nan count", int(torch.isnan(out.float()).sum()), "min ref where nan", ref[torch.isnan(out.float())].min().item() if torch.isnan(out.float()).any() else None,
      "max ref where nan", ref[torch.isnan(out.float())].max().item() if torch.isnan(out.float()).any() else None)
pre = x.float() @ w.float().t() + (b if BIAS else 0)
if torch.isnan(out.float()).any():
    print("pre-activation at nan positions: min", pre[torch.isnan(out.float())].min().item(), "max", pre[torch.isnan(out.float())].max().item())
err = (out.float() - ref).abs()
bad = (err > 0.05 * (1 + ref.abs())) | torch.isnan(err)
print("bad elements", int(bad.sum()), "of", M * N)
mb = bad.float().reshape(M // 16, 16, N // 16, 16).sum((1, 3)).cpu()
for r in range(min(M // 16, 32)):
    print("".join("#" if v == 256 else ("+" if v > 0 else ".") for v in mb[r].tolist()[:64]))

idx = bad.nonzero()[:24].cpu()
for r, c in idx.tolist():
    print(f"row {r:4d} col {c:4d}  out {out[r, c].item():9.4f}  ref {ref[r, c].item():9.4f}  pre {pre[r, c].item():9.4f}")
