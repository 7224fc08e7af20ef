"""Two half-batch chains on two HIP streams against one full-batch chain (GPU box only).

The shared layers' GEMMs are persistent 1-workgroup-per-CU launches whose epilogues (fp32 read-modify-write of the residual
stream, packed bf16 stores) all fall on the same instants: during them nobody issues MFMAs, and during the main loops the
HBM idles.  Splitting the batch into two independent chains (samples are independent) and capping each chain's GEMM at
half the CUs lets one chain's memory-bound phases (epilogues, LayerNorm, attention) run beside the other chain's MFMA phases.
This probe runs N transformer layers of the C2 shapes both ways with the product kernels and prints the time per layer.

    python tools/probes/two_chain_probe.py [--layers 11] [--caps 128 144 160]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip  # noqa: E402

D, H, LV, LT = 768, 12, 50, 77


def make_chain(nimg, ntxt, g):
    Mv, M = nimg * LV, nimg * LV + ntxt * LT
    dev = "cuda"
    w = dict(Mv=Mv, M=M, nimg=nimg, ntxt=ntxt)
    w["X"] = torch.randn(M, D, generator=g).to(dev)
    w["LNO"] = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    w["QKV"] = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
    w["AO"] = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    w["HID"] = torch.empty(M, 4 * D, dtype=torch.bfloat16, device=dev)
    return w


FOLD = False


def layer_fold(w, P):
    """The shipped layer loop's form: LayerNorms folded into the GEMMs (producer epilogues on out_proj / c_proj, consumer on
    in_proj / c_fc, rowstat_finalize between them); M must be a multiple of 256."""
    X, LNO, QKV, AO, HID, Mv, M = w["X"], w["LNO"], w["QKV"], w["AO"], w["HID"], w["Mv"], w["M"]
    if "CEN" not in w:
        w["CEN"] = torch.zeros(M, device="cuda")
        w["RST"] = torch.stack([torch.ones(M), torch.zeros(M)], 1).cuda().contiguous()
        w["PART"] = torch.empty(M, D // 64, 2, device="cuda")
    CEN, RST, PART = w["CEN"], w["RST"], w["PART"]
    hip.gemm(LNO, P["wqkv"], QKV, bias=P["bqkv"], fold_in=hip.FoldIn(RST, P["cqkv"]))
    hip.attention(QKV[:Mv], AO[:Mv], w["nimg"], LV, H, False)
    hip.attention(QKV[Mv:], AO[Mv:], w["ntxt"], LT, H, True)
    hip.gemm(AO, P["wo"], X, bias=P["bo"], resid=X, resid_kind=hip.RESID_F32, fold_out=hip.FoldOut(LNO, CEN, PART))
    hip.rowstat_finalize(PART, CEN, RST, M, D)
    hip.gemm(LNO, P["wfc"], HID, bias=P["bfc"], act=hip.ACT_QUICKGELU, fold_in=hip.FoldIn(RST, P["cfc"]))
    hip.gemm(HID, P["wpr"], X, bias=P["bpr"], resid=X, resid_kind=hip.RESID_F32, fold_out=hip.FoldOut(LNO, CEN, PART))
    hip.rowstat_finalize(PART, CEN, RST, M, D)


def layer(w, P):
    if FOLD:
        return layer_fold(w, P)
    X, LNO, QKV, AO, HID, Mv, M = w["X"], w["LNO"], w["QKV"], w["AO"], w["HID"], w["Mv"], w["M"]
    hip.layernorm_split(X, P["g"], P["b"], P["g"], P["b"], Mv, LNO, M)
    hip.gemm(LNO, P["wqkv"], QKV, bias=P["bqkv"])
    hip.attention(QKV[:Mv], AO[:Mv], w["nimg"], LV, H, False)
    hip.attention(QKV[Mv:], AO[Mv:], w["ntxt"], LT, H, True)
    hip.gemm(AO, P["wo"], X, bias=P["bo"], resid=X, resid_kind=hip.RESID_F32)
    hip.layernorm_split(X, P["g"], P["b"], P["g"], P["b"], Mv, LNO, M)
    hip.gemm(LNO, P["wfc"], HID, bias=P["bfc"], act=hip.ACT_QUICKGELU)
    hip.gemm(HID, P["wpr"], X, bias=P["bpr"], resid=X, resid_kind=hip.RESID_F32)


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=11)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--caps", type=int, nargs="+", default=[128, 144, 160, 192, 0])
    ap.add_argument("--mode", default="all", help="all | full | two (profiling: one schedule only, first cap)")
    ap.add_argument("--split", type=float, default=0.5, help="share of the samples in chain A")
    ap.add_argument("--delays", type=float, nargs="+", default=[0.0], help="chain B starts this many us late (a LayerNorm over scratch rows)")
    ap.add_argument("--fold", action="store_true", help="the LayerNorm-fold layer form (batch a multiple of 512: whole 256-row tiles per half)")
    ap.add_argument("--lt", type=int, default=77, help="rows per caption (34 ~ packed captions)")
    args = ap.parse_args()
    FOLD = args.fold
    LT = args.lt
    g = torch.Generator().manual_seed(0)
    P = dict(g=torch.ones(D).cuda(), b=torch.zeros(D).cuda())
    for name, n, k in (("qkv", 3 * D, D), ("o", D, D), ("fc", 4 * D, D), ("pr", D, 4 * D)):
        P["w" + name] = (torch.randn(n, k, generator=g) * 0.02).to(torch.bfloat16).cuda()
        P["b" + name] = (torch.randn(n, generator=g) * 0.02).cuda()
        P["c" + name] = P["w" + name].float().sum(1).contiguous()
    B = args.batch
    full = make_chain(B, B, g)
    na = int(round(B * args.split / 4)) * 4
    halves = [make_chain(na, na, g), make_chain(B - na, B - na, g)]
    scratch = torch.randn(65024, D, device="cuda")
    scratch_o = torch.empty(65024, D, dtype=torch.bfloat16, device="cuda")
    cur = torch.cuda.current_stream()
    hi = hip.priority_stream(torch.device("cuda", 0), True)
    streams = [hi, hip.priority_stream(torch.device("cuda", 0), True)]

    def run_full():
        for _ in range(args.layers):
            layer(full, P)

    def run_serial_halves():
        for _ in range(args.layers):
            for h in halves:
                layer(h, P)

    def run_two(cap, delay=0.0):
        drows = int(65024 * delay / 47.0)

        def fn():
            ev = torch.cuda.Event()
            ev.record(cur)
            prev = hip.set_wg_cap(cap)
            for st in streams:
                st.wait_event(ev)
            if drows:
                with torch.cuda.stream(streams[1]):
                    hip.layernorm_split(scratch, P["g"], P["b"], P["g"], P["b"], drows, scratch_o, drows)
            # interleave the launches of the two chains layer by layer (one host thread feeds both queues)
            for _ in range(args.layers):
                for st, h in zip(streams, halves):
                    with torch.cuda.stream(st):
                        layer(h, P)
            for st in streams:
                cur.wait_stream(st)
            hip.set_wg_cap(prev)
        return fn

    if args.mode == "full":
        print(f"one chain: {timed(run_full, args.iters):.3f} ms")
        sys.exit(0)
    if args.mode == "two":
        print(f"two chains cap {args.caps[0]}: {timed(run_two(args.caps[0]), args.iters):.3f} ms")
        sys.exit(0)
    t_full = timed(run_full, args.iters)
    print(f"one chain, batch {B}: {t_full:.3f} ms = {t_full / args.layers * 1e3:.1f} us per layer")
    t_ser = timed(run_serial_halves, args.iters)
    print(f"two half chains back to back on one stream: {t_ser:.3f} ms = {t_ser / args.layers * 1e3:.1f} us per layer")
    for cap in args.caps:
        for dl in args.delays:
            t = timed(run_two(cap, dl), args.iters)
            print(f"two chains on two streams, wg_cap {cap}, delay {dl} us: {t:.3f} ms = {t / args.layers * 1e3:.1f} us per layer ({t_full / t:.3f} x)")
    t_full = timed(run_full, args.iters)
    print(f"one chain again: {t_full:.3f} ms")
