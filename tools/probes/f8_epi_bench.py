"""Isolated launch times of the fp8 (e4m3, v_mfma_scale_f32_16x16x128_f8f6f4) projections of BASELINE config C5 (GPU box only):
c_fc (e4m3 output, QuickGELU) and c_proj (fp32 residual update, plain and as the LayerNorm-fold producer) at the L/14 step's row
count, on the product library or on a probe build (MSCLIP_HIP_LIB=tools/probes/libgemm_noepi.so: tiles skip their epilogue)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip

M, D = int(os.environ.get("F8_M", "74752")), 1024      # 256 x 257 image rows + the packed caption rows of batch 256, whole tiles
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()


def t(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


res = {}
xq, xs = hip.quantize_rows_f8(r(M, D)); xq, xs = xq.cuda(), xs.cuda()
wfc, sfc = hip.quantize_rows_f8(r(4 * D, D, sc=0.03)); bfc = r(4 * D)
hq = torch.empty(M, 4 * D, dtype=torch.uint8, device="cuda")
res["c_fc f8 -> e4m3 hidden"] = t(lambda: hip.gemm_f8(xq, wfc, hq, xs, sfc, bias=bfc, act=hip.ACT_QUICKGELU, out_scale=8.0))
hb = torch.empty(M, 4 * D, dtype=torch.bfloat16, device="cuda")
res["c_fc f8 -> bf16 hidden"] = t(lambda: hip.gemm_f8(xq, wfc, hb, xs, sfc, bias=bfc, act=hip.ACT_QUICKGELU))
wpr, spr = hip.quantize_rows_f8(r(D, 4 * D, sc=0.03)); bpr = r(D)
hs = torch.full((M,), 0.125, device="cuda")
X = r(M, D)
res["c_proj f8 plain"] = t(lambda: hip.gemm_f8(hq, wpr, X, hs, spr, bias=bpr, resid=X, resid_kind=hip.RESID_F32))
xb, cen, part = torch.empty(M, D, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, device="cuda"), torch.empty(M, D // 64, 2, device="cuda")
res["c_proj f8 producer"] = t(lambda: hip.gemm_f8(hq, wpr, X, hs, spr, bias=bpr, resid=X, resid_kind=hip.RESID_F32,
                                                 fold_out=hip.FoldOut(xb, cen, part)))
wq, sq = hip.quantize_rows_f8(r(3 * D, D, sc=0.03)); bq = r(3 * D)
qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device="cuda")
res["in_proj f8 (fp8-qkv)"] = t(lambda: hip.gemm_f8(xq, wq, qkv, xs, sq, bias=bq))
flops = {"c_fc": 2.0 * M * 4 * D * D, "c_pr": 2.0 * M * 4 * D * D, "in_p": 2.0 * M * 3 * D * D}
print("JSON", json.dumps({"M": M, "lib": os.environ.get("MSCLIP_HIP_LIB") or "product", "us": {k: round(v, 1) for k, v in res.items()},
                          "tflops": {k: round(flops[k[:4]] / v / 1e6, 1) for k, v in res.items()}}))
