"""Fused front passes against the unfused chains they replace (batch 512, 224x224)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msclip_amd import hip, packing as P

BF = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B, S = 512, 224
    g = torch.Generator().manual_seed(0)
    img = torch.randn(B, 3, S, S, generator=g).cuda()
    w, b = (torch.randn(27, 96, generator=g) * 0.3).cuda(), (torch.randn(96, generator=g) * 0.2).cuda()
    Hm = S // 2
    for cout in (96,):
        c2 = P.ConvSpec(torch.randn(cout, 48, 3, 3, generator=g) * 0.07, torch.randn(cout, generator=g) * 0.2, Hm, Hm, 2, 1).to("cuda")
        Ho = c2.h_out
        oa = torch.empty(B * Hm * Hm, 48, dtype=BF, device="cuda")
        ob = torch.empty(B * Hm * Hm, 48, dtype=BF, device="cuda")
        out = torch.empty(B * Ho * Ho, cout, dtype=BF, device="cuda")
        t_a = timeit(lambda: hip.stem_conv_dual(img, w, b, oa, ob))
        t_b = timeit(lambda: hip.gemm(oa, c2.weight, out, M=B * Ho * Ho, N=cout, bias=c2.bias, act=hip.ACT_RELU,
                                      conv=c2.geometry(), ktab=c2.ktab))
        t_f = timeit(lambda: hip.stem_dual_conv3x3s2(img, w, b, ob, c2.weight, c2.bias, out))
        print(f"stem  Cout={cout}: unfused {t_a:.1f} + {t_b:.1f} = {t_a + t_b:.1f} us   fused {t_f:.1f} us", flush=True)
    xs = torch.zeros(B * Hm * Hm * 48 + 64, dtype=BF, device="cuda")      # the dense K-padded loader reads 16 past a row
    x = xs[:B * Hm * Hm * 48].view(B, Hm, Hm, 48)
    x.copy_(torch.randn(B, Hm, Hm, 48, generator=g).to(BF))
    c1 = P.ConvSpec(torch.randn(48, 48, 1, 1, generator=g) * 0.2, torch.randn(48, generator=g) * 0.2, Hm, Hm, 1, 0).to("cuda")
    c2 = P.ConvSpec(torch.randn(48, 48, 3, 3, generator=g) * 0.07, torch.randn(48, generator=g) * 0.2, Hm, Hm, 2, 1).to("cuda")
    Ho = c2.h_out
    t1 = torch.empty(B * Hm * Hm + 8, 48, dtype=BF, device="cuda")[:B * Hm * Hm]
    out = torch.empty(B * Ho * Ho, 48, dtype=BF, device="cuda")
    t_a = timeit(lambda: hip.gemm(x.view(-1, 48), c1.weight, t1, M=B * Hm * Hm, N=48, bias=c1.bias, act=hip.ACT_RELU, ldx=48))
    t_b = timeit(lambda: hip.gemm(t1, c2.weight, out, M=B * Ho * Ho, N=48, bias=c2.bias, act=hip.ACT_RELU,
                                  conv=c2.geometry(), ktab=c2.ktab))
    t_f = timeit(lambda: hip.conv1x1_conv3x3s2(x, c1.weight, c1.bias, c2.weight, c2.bias, out, B, Hm, Hm))
    print(f"par1  Cout=48: unfused {t_a:.1f} + {t_b:.1f} = {t_a + t_b:.1f} us   fused {t_f:.1f} us", flush=True)
    cr = P.ConvSpec(torch.randn(96, 48, 1, 1, generator=g) * 0.14, torch.randn(96, generator=g) * 0.2, Hm, Hm, 2, 0).to("cuda")
    c3 = P.ConvSpec(torch.randn(96, 48, 1, 1, generator=g) * 0.14, torch.randn(96, generator=g) * 0.2, Ho, Ho, 1, 0).to("cuda")
    tr = torch.empty(B * Ho * Ho, 96, dtype=BF, device="cuda")
    o3 = torch.empty(B * Ho * Ho, 96, dtype=BF, device="cuda")
    outs = torch.zeros(B * Ho * Ho * 48 + 64, dtype=BF, device="cuda")[:B * Ho * Ho * 48].view(-1, 48)
    t_c = timeit(lambda: hip.gemm(x, cr.weight, tr, M=B * Ho * Ho, N=96, bias=cr.bias, conv=cr.geometry(), ktab=cr.ktab))
    t_d = timeit(lambda: hip.gemm(outs, c3.weight, o3, M=B * Ho * Ho, N=96, bias=c3.bias, act=hip.ACT_RELU, resid=tr,
                                  resid_kind=hip.RESID_BF16, ldx=48))
    b3r = (c3.bias + cr.bias).contiguous()
    t_g = timeit(lambda: hip.convresblock48_s2(x, c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, cr.weight, b3r, o3, B, Hm, Hm))
    print(f"block: fused conv1-conv2 {t_f:.1f} + shortcut {t_c:.1f} + conv3 {t_d:.1f} = {t_f + t_c + t_d:.1f} us   one launch {t_g:.1f} us", flush=True)


if __name__ == "__main__":
    main()
