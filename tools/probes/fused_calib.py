"""Calibration for the fused QKV-projection + attention kernel (GPU box only): what the parts cost today at the packed C2 shapes --
the in_proj GEMM on the ping-pong kernel (256 x 256 tiles) and on 256 x 192 two-buffer tiles (one head's q|k|v per tile: the
main loop a fused kernel would run), and the attention kernels on the same rows."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip, synth

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


res = {}
D, H = 768, 12
w, b = r(3 * D, D, sc=0.03, dt=BF), r(3 * D)
for M in (25600, 17664, 43264):
    x = r(M, D, dt=BF)
    out = torch.empty(M, 3 * D, dtype=BF, device="cuda")
    for tile in (4, 6):
        res[f"qkv M={M} tile{tile}"] = t(lambda: hip.gemm(x, w, out, bias=b, tile=tile))
    o4 = torch.empty_like(out); hip.gemm(x, w, o4, bias=b, tile=4)
    o6 = torch.empty_like(out); hip.gemm(x, w, o6, bias=b, tile=6)
    res[f"qkv M={M} tile6 vs tile4 max abs diff"] = (o4.float() - o6.float()).abs().max().item()
Bi, Lv = 512, 50
qkv = r(Bi * Lv, 3 * D, dt=BF); ao = torch.empty(Bi * Lv, D, dtype=BF, device="cuda")
res["attention image 512 x 50"] = t(lambda: hip.attention(qkv, ao, Bi, Lv, H, False))
tok = synth.synth_tokens(512, seed=6).cuda()
length = torch.empty(512, dtype=torch.int32, device="cuda"); cu = torch.empty(514, dtype=torch.int32, device="cuda")
hip.text_lengths(tok, length, cu)
total, lmax = int(cu[512]), int(cu[513])
Mt = (total + 255) // 256 * 256
qkv_t = r(Mt, 3 * D, dt=BF); ao_t = torch.empty(Mt, D, dtype=BF, device="cuda")
res[f"attention text packed ({total} rows, Lmax {lmax})"] = t(lambda: hip.attention_varlen(qkv_t, ao_t, cu, 512, lmax, H, True, pad_rows=Mt - total))
for k, v in res.items():
    print(f"{k:60s} {v:10.2f}")
print("JSON", json.dumps({k: round(v, 2) for k, v in res.items()}))
