"""Attention backward at the packed C2 shapes (GPU box only): us per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip, synth
BF = torch.bfloat16
H, D = 12, 768
def t(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
Bi, Lv = 512, 50
qkv = torch.randn(Bi * Lv, 3 * D, device="cuda").to(BF); o = torch.randn(Bi * Lv, D, device="cuda").to(BF); do = torch.randn_like(o); dq = torch.empty_like(qkv)
print("attention_bwd image 512 x 50: %.1f us" % t(lambda: hip.attention_bwd(qkv, o, do, dq, Bi, Lv, H, False)))
tok = synth.synth_tokens(512, seed=6).cuda()
length = torch.empty(512, dtype=torch.int32, device="cuda"); cu = torch.empty(514, dtype=torch.int32, device="cuda")
hip.text_lengths(tok, length, cu)
total, lmax = int(cu[512]), int(cu[513]); Mt = (total + 255) // 256 * 256
qkv = torch.randn(Mt, 3 * D, device="cuda").to(BF); o = torch.randn(Mt, D, device="cuda").to(BF); do = torch.randn_like(o); dq = torch.empty_like(qkv)
print("attention_bwd captions packed (%d rows, Lmax %d): %.1f us" % (total, lmax, t(lambda: hip.attention_bwd_varlen(qkv, o, do, dq, cu, 512, lmax, H, True, pad_rows=Mt - total))))
Bt, Lt = 512, 77
qkv = torch.randn(Bt * Lt, 3 * D, device="cuda").to(BF); o = torch.randn(Bt * Lt, D, device="cuda").to(BF); do = torch.randn_like(o); dq = torch.empty_like(qkv)
print("attention_bwd captions 512 x 77 (full rows): %.1f us" % t(lambda: hip.attention_bwd(qkv, o, do, dq, Bt, Lt, H, True)))
