"""Train-mode BatchNorm backward of a residual block's two BatchNorms behind one ReLU: msclip_relu_bwd + a reduce / dx pass pair
per BatchNorm (round 5) against msclip_bn_bwd_fused (mask on the fly, both BatchNorms per pass, 4 columns per thread)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip  # noqa: E402

BF = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    B = 512
    for hw, C in ((112, 48), (56, 96), (28, 192), (14, 384), (7, 768)):
        M = B * hw * hw
        g = torch.Generator().manual_seed(0)
        dy = torch.randn(M, C, generator=g).to(BF).cuda()
        dy2 = torch.randn(M, C, generator=g).to(BF).cuda()
        y = torch.relu(torch.randn(M, C, generator=g)).to(BF).cuda()
        xs = [torch.randn(M, C, generator=g).cuda() for _ in range(2)]
        gam = torch.ones(C, device="cuda")
        st = [hip.bn_stats(x, gamma=gam, beta=torch.zeros_like(gam)) for x in xs]
        dxs = [torch.empty(M, C, dtype=BF, device="cuda") for _ in xs]
        dpre = torch.empty(M, C, dtype=BF, device="cuda")

        def old():
            hip.relu_bwd(dy, y, dpre, dy2=dy2)
            for x, s, dx in zip(xs, st, dxs):
                hip.bn_bwd(dpre, x, s[0], s[2], gam, dx)

        def new():
            hip.bn_bwd_fused(dy, [(x, s[0], s[2], gam, dx) for x, s, dx in zip(xs, st, dxs)], y=y, dy2=dy2)

        def new1():
            hip.bn_bwd_fused(dy, [(xs[0], st[0][0], st[0][2], gam, dxs[0])], y=y)

        def old1():
            hip.relu_bwd(dy, y, dpre)
            hip.bn_bwd(dpre, xs[0], st[0][0], st[0][2], gam, dxs[0])

        if os.environ.get("SWEEP"):
            r = 1
            while C * r < 768 and M % (r * 2) == 0 and M // (r * 2) >= 64:
                r *= 2
            Mw = M // r
            sides = [(x, s[0], s[2], gam, dx) for x, s, dx in zip(xs, st, dxs)]
            row = []
            for rc in (32, 64, 128, 256, 512, 1024):
                for dc in (16, 32, 64, 128):
                    ch = max(1, min(2048, Mw // rc))
                    t = timeit(lambda: hip.bn_bwd_fused(dy, sides, y=y, dy2=dy2, chunks=ch, dx_chunks=max(1, Mw // dc)), n=10)
                    row.append((t, rc, dc, ch))
            row.sort()
            print(f"{hw:4d}^2 x {C:3d} (Mw {Mw}): best (us, reduce rows/chunk, dx rows/chunk, chunks)", [(round(t, 1), a, b, c) for t, a, b, c in row[:5]],
                  "worst", round(row[-1][0], 1), flush=True)
            continue
        e = M * C
        t_old, t_new, t_old1, t_new1 = timeit(old), timeit(new), timeit(old1), timeit(new1)
        b_old, b_new = e * (8 + 2 * (6 + 8)), e * (14 + 14 + 4)
        print(f"{hw:4d}^2 x {C:3d}: two BNs + dy2: old {t_old:8.1f} us ({b_old / t_old / 1e6:5.2f} TB/s)  fused {t_new:8.1f} us "
              f"({b_new / t_new / 1e6:5.2f} TB/s of its bytes) | one BN: old {t_old1:8.1f}  fused {t_new1:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
