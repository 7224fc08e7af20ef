"""Micro-benchmark of the GEMM kernel on the transformer's projection shapes (GPU box only).
    python tools/gemm_bench.py [--tiles 1 2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msclip_amd import hip  # noqa: E402

SHAPES = [  # (name, M, N, K, epilogue)
    ("qkv  ", 65024, 2304, 768, "bias"),
    ("out  ", 65024, 768, 768, "resid"),
    ("fc   ", 65024, 3072, 768, "gelu"),
    ("proj ", 65024, 768, 3072, "resid"),
    ("txt0 ", 39424, 3072, 768, "gelu"),
    ("qkvnb", 65024, 2304, 768, "none"),
    ("qkvs ", 2048, 2304, 768, "bias"),        # 72 tiles: a quarter of the CUs busy (epilogue without chip-wide bursts)
    ("fcs  ", 2048, 3072, 768, "gelu"),
    ("outs ", 6144, 768, 768, "resid"),
]
FULL = False
ZEROS = False
SQUARE = [("sq8k ", 8192, 8192, 8192, "bias"), ("sq4k ", 4096, 4096, 4096, "bias")]   # calibration against published tiers


def run(name, M, N, K, epi, tile, iters=20, check=False):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(M, K, generator=g)).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).cuda()
    if ZEROS:                                       # power calibration: all-zero operands
        x.zero_(); w.zero_()
    b = torch.randn(N, generator=g).cuda()
    if epi == "resid":
        out = torch.randn(M, N, generator=g).cuda()
        kw = dict(bias=b, resid=out, resid_kind=hip.RESID_F32)
    elif epi == "gelu":
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        kw = dict(bias=b, act=hip.ACT_QUICKGELU)
    elif epi == "none":
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        kw = dict()
        b = torch.zeros_like(b)
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        kw = dict(bias=b)
    if check:
        ref_in = out.clone() if epi == "resid" else None
        hip.gemm(x, w, out, tile=tile, **kw)
        rows = torch.tensor([0, 1, 255, 256, 4095, M - 1, M // 2 + 3], device="cuda").clamp(max=M - 1)
        ref = x[rows].float() @ w.float().t() + b
        if epi == "gelu":
            ref = ref * torch.sigmoid(1.702 * ref)
        if epi == "resid":
            ref = ref + ref_in[rows]
        err = (out[rows].float() - ref).abs().max().item()
        if not os.environ.get("MSCLIP_HIP_LIB"):
            assert err < 0.05 * max(1.0, ref.abs().max().item()), (name, tile, err)
        if FULL:                                    # every output element, three launches (race screen)
            for rep in range(3):
                if epi == "resid":
                    out.copy_(ref_in)
                hip.gemm(x, w, out, tile=tile, **kw)
                worst = 0.0
                for r0 in range(0, M, 8192):
                    r = x[r0:r0 + 8192].float() @ w.float().t() + b
                    if epi == "gelu":
                        r = r * torch.sigmoid(1.702 * r)
                    if epi == "resid":
                        r = r + ref_in[r0:r0 + 8192]
                    d = (out[r0:r0 + 8192].float() - r).abs()
                    tol = 0.02 * r.abs() + 0.02 * r.abs().mean()
                    bad = (d > tol).sum().item()
                    worst = max(worst, d.max().item())
                    assert bad == 0, (name, tile, rep, r0, bad, worst)
            print(f"   full check {name} tile{tile}: ok (max abs err {worst:.4f})")
    for _ in range(3):
        hip.gemm(x, w, out, tile=tile, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        hip.gemm(x, w, out, tile=tile, **kw)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--square", action="store_true")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--zeros", action="store_true")
    args = ap.parse_args()
    FULL = args.full
    ZEROS = args.zeros
    if args.square:
        SHAPES = SQUARE
    tot = {t: 0.0 for t in args.tiles}
    for name, M, N, K, epi in SHAPES:
        line = f"{name} M={M} N={N} K={K} {epi:5s}"
        for t in args.tiles:
            if t == 7 and epi == "resid":           # the 4-wave kernel takes bf16 outputs without residual only
                line += " | tile7:      n/a"
                continue
            us, tf = run(name, M, N, K, epi, t, check=True)
            if name.strip() in ("qkv", "out", "fc", "proj"):
                tot[t] += us
            line += f" | tile{t}: {us:8.1f} us {tf:7.1f} TF"
        print(line)
    print("sum of the 4 shared-layer GEMMs (us):", {t: round(v, 1) for t, v in tot.items()})
