"""The fused in_proj + attention kernel (msclip_qkv_attention) against the launches it replaces, at the packed C2 shapes (GPU box only)."""
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


D, H, Bi, Lv, Bt = 768, 12, 512, 50, 512
tok = synth.synth_tokens(Bt, seed=6).cuda()
length = torch.empty(Bt, dtype=torch.int32, device="cuda"); cut = torch.empty(Bt + 2, dtype=torch.int32, device="cuda")
hip.text_lengths(tok, length, cut)
total, lmax = int(cut[Bt]), int(cut[Bt + 1])
Mv = Bi * Lv
live = Mv + total
M = Mv + (total + 255) // 256 * 256
cu = torch.cat([torch.arange(0, Mv, Lv, dtype=torch.int32, device="cuda"), cut[:Bt] + Mv, torch.tensor([live], dtype=torch.int32, device="cuda")]).contiguous()
x = r(M, D, dt=BF)
w, b = r(3 * D, D, sc=0.03, dt=BF), r(3 * D)
wh, bh = hip.head_major_qkv(w, b, H)
qkv = torch.empty(M, 3 * D, dtype=BF, device="cuda")
ao, ao2 = torch.empty(M, D, dtype=BF, device="cuda"), torch.zeros(M, D, dtype=BF, device="cuda")
tabs = hip.QkvAttnTables(cu, Bi + Bt, split_sample=Bi, total_rows=live)
res = {"rows": M, "live_rows": live, "tiles": int(tabs.ntiles)}
res["in_proj (ping-pong GEMM)"] = t(lambda: hip.gemm(x, w, qkv, bias=b))
res["attention image"] = t(lambda: hip.attention(qkv[:Mv], ao[:Mv], Bi, Lv, H, False))
res["attention captions"] = t(lambda: hip.attention_varlen(qkv[Mv:], ao[Mv:], cut, Bt, lmax, H, True, pad_rows=M - live))
res["three launches"] = res["in_proj (ping-pong GEMM)"] + res["attention image"] + res["attention captions"]
res["fused msclip_qkv_attention"] = t(lambda: hip.qkv_attention(x, wh, bh, ao2, tabs, H, causal_from_row=Mv, M=M))
res["in_proj on 256 x 192 two-buffer tiles (tile 6), no attention"] = t(lambda: hip.gemm(x, w, qkv, bias=b, tile=6))
d = (ao2[:live].float() - ao[:live].float()).abs()
res["max abs diff fused vs chain"] = float(d.max()); res["mean abs diff"] = float(d.mean()); res["ref abs mean"] = float(ao[:live].float().abs().mean())
for k, v in res.items():
    print(f"{k:64s} {v:12.4f}" if isinstance(v, float) else f"{k:64s} {v}")
print("JSON", json.dumps(res))
