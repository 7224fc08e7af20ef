"""Parity-class dgrad of the 3x3 / stride-2 convolutions (train_conv.ConvSideBackward._dgrad_parity) against fp32 torch and against
the column-matrix path (dcol GEMM + col2im), per geometry of the B/32 conv side at batch 512."""
import os
import sys
import time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from msclip_amd import hip, packing as P, train_conv as TC                 # noqa: E402
BF = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = torch.Generator().manual_seed(3)
bw = object.__new__(TC.ConvSideBackward)
bw._wt, bw._pplan = {}, TC._PARITY_PLANS
for name, ci, co, h in [("par1.conv2", 48, 48, 112), ("stem0.conv1", 48, 96, 112), ("par2.conv2", 96, 96, 56), ("stem1.conv1", 96, 192, 56),
                        ("par3.conv2", 192, 192, 28), ("stem2.conv1", 192, 384, 28)]:
    w = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5
    spec = P.ConvSpec(w, torch.zeros(co), h, h, 2, 1).to("cuda")
    ho = spec.h_out
    pix = B * ho * ho
    dpre = TC._zbuf(pix, co, "cuda")
    dpre.copy_((torch.randn(pix, co, generator=g) * 0.5).to(BF))
    x_in = TC._zbuf(B * h * h, ci, "cuda")
    x_in.zero_()

    # dgrad only (the weight gradient is not part of this probe)
    def dgrad_new():
        dx = TC._zbuf(B * h * h, ci, "cuda")
        bw._dgrad_parity(bw._parity_plan(name, spec), spec, dpre, dx, B)
        return dx

    def dgrad_old():
        kp = spec.weight.shape[1]
        wt = bw._w_t(name, spec.weight, co, kp)
        dcol = torch.empty(pix, kp, dtype=BF, device="cuda")
        hip.gemm(dpre, wt, dcol, M=pix, N=kp, ldx=co)
        dx = TC._zbuf(B * h * h, ci, "cuda")
        hip.col2im(dcol, dx, B, h, h, ci, 3, 3, 2, 1)
        return dx
    assert bw._parity_ok(spec) and bw._parity_plan(name, spec), name
    a, o = dgrad_new(), dgrad_old()
    # fp32 reference on a few samples: conv_transpose of the bf16 operands
    nb = min(B, 4)
    wq = spec.weight[:, :9 * ci].float().view(co, 3, 3, ci).permute(0, 3, 1, 2)             # [co, ci, 3, 3] as packed (bf16 values)
    dy = dpre[:nb * ho * ho].float().view(nb, ho, ho, co).permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(dy, wq, stride=2, padding=1, output_padding=1).permute(0, 2, 3, 1).reshape(nb * h * h, ci)
    sc = ref.abs().max().item()
    ea, eo = (a[:nb * h * h].float() - ref).abs().max().item() / sc, (o[:nb * h * h].float() - ref).abs().max().item() / sc
    full = (a.float() - o.float()).abs().max().item() / sc

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 5 * 1e6
    print(f"{name:12s} {ci:3d}->{co:3d} @{h:3d}: rel err new {ea:.2e} old {eo:.2e}, new vs old over the batch {full:.2e}; "
          f"parity {timed(dgrad_new):7.0f} us, dcol + col2im {timed(dgrad_old):7.0f} us")
