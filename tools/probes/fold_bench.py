"""Isolated launch times of the LayerNorm-fold epilogue forms against the standard ones (GPU box only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip

M, D = int(os.environ.get("FOLD_M", "65024")), 768
BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0, dt=torch.float32: (torch.randn(*s, generator=g) * sc).to(dt).cuda()


def t(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


x = r(M, D, dt=BF)
rstat = torch.stack([1.0 + 0.1 * torch.rand(M), 0.1 * torch.randn(M)], 1).cuda().contiguous()
res = {}
for name, N, act in (("qkv", 2304, 0), ("fc", 3072, 1)):
    w, w2, b, b2 = r(N, D, sc=0.03, dt=BF), r(N, D, sc=0.03, dt=BF), r(N), r(N)
    c, c2 = w.float().sum(1).contiguous(), w2.float().sum(1).contiguous()
    out = torch.empty(M, N, dtype=BF, device="cuda")
    for rep in range(2):
        res[name + " plain"] = t(lambda: hip.gemm(x, w, out, bias=b, act=act))
        res[name + " fold 1seg"] = t(lambda: hip.gemm(x, w, out, bias=b, act=act, fold_in=hip.FoldIn(rstat, c)))
        res[name + " fold 2seg"] = t(lambda: hip.gemm(x, w, out, bias=b, act=act, fold_in=hip.FoldIn(rstat, c, w2, b2, c2, 25600)))
        print(rep, {k: round(v, 1) for k, v in res.items() if k.startswith(name)})
for name, K in (("out", 768), ("proj", 3072)):
    a, w, b = r(M, K, dt=BF), r(D, K, sc=0.03, dt=BF), r(D)
    X = r(M, D)
    xb, cen, part = torch.empty(M, D, dtype=BF, device="cuda"), torch.zeros(M, device="cuda"), torch.empty(M, D // 64, 2, device="cuda")
    for rep in range(2):
        res[name + " plain"] = t(lambda: hip.gemm(a, w, X, bias=b, resid=X, resid_kind=hip.RESID_F32))
        res[name + " producer"] = t(lambda: hip.gemm(a, w, X, bias=b, resid=X, resid_kind=hip.RESID_F32, fold_out=hip.FoldOut(xb, cen, part)))
        print(rep, {k: round(v, 1) for k, v in res.items() if k.startswith(name)})
rs = torch.empty(M, 2, device="cuda")
print("finalize", round(t(lambda: hip.rowstat_finalize(part, cen, rs, M, D)), 1), "us;  ln_pair",
      round(t(lambda: hip.layernorm_split(X, b, b, b, b, 25600, xb, M)), 1), "us")
# one line per launch form for profiles/: name, us, TFLOP/s
import json
flops = {"qkv": 2.0 * M * 2304 * D, "fc": 2.0 * M * 3072 * D, "out": 2.0 * M * D * D, "proj": 2.0 * M * D * 3072}
print("JSON", json.dumps({"M": M, "lib": os.environ.get("MSCLIP_HIP_LIB") or "product",
                          "us": {k: round(v, 1) for k, v in res.items()},
                          "tflops": {k: round(flops[k.split()[0]] / v / 1e6, 1) for k, v in res.items()}}))
