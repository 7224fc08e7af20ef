"""Per-launch timing of the implicit-conv GEMMs inside one forward (GPU box only)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msclip_amd import hip, synth
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
name = "b32-yfcc-msclips"
schema = [(k, tuple(s), getattr(torch, d)) for k, s, d in json.load(open(f"tests/golden/{name}.schema.json"))]
m = get_clip_model(named_config(name)); m.load_state_dict(synth.synth_state_dict(schema)); m = m.cuda().eval()
B = 512
img = synth.synth_images(B).cuda(); tok = synth.synth_tokens(B).cuda()
eng = m.engine()
for _ in range(2): eng.run(img, tok)
pr = hip.KernelProbe(); hip.set_gemm_probe(1, pr)
pr0 = hip.KernelProbe(); hip.set_gemm_probe(0, pr0)
eng.run(img, tok); torch.cuda.synchronize(); hip.set_gemm_probe(1, None); hip.set_gemm_probe(0, None)
tot = 0
for (s, e, fl), tag in zip(pr.records, pr.tags):
    us = s.elapsed_time(e) * 1e3; tot += us
    M, N, K, kalg, conv = tag
    H, W, Cin, Ho, Wo, st, pad = conv
    inb = B * H * W * Cin * 2 / 1e6; outb = M * N * 2 / 1e6
    print(f"M={M:8d} N={N:4d} Kpad={K:5d} Kalg={kalg:5d} in {H:3d}x{W:<3d}x{Cin:<4d} s{st} -> {Ho:3d}: {us:8.1f} us  {fl/us/1e6:7.1f} TF  min-traffic {inb+outb:7.1f} MB -> {(inb+outb)/us*1e6/1e6:6.2f} TB/s")
print("total conv-gemm us", tot)

import collections
agg = collections.OrderedDict()
for (s_, e_, fl), tag in zip(pr0.records, pr0.tags):
    us = s_.elapsed_time(e_) * 1e3
    key = tag[:3]
    a = agg.setdefault(key, [0, 0.0, fl]); a[0] += 1; a[1] += us
tot0 = 0
for (M, N, K), (n, us, fl) in agg.items():
    tot0 += us
    print(f"dense M={M:8d} N={N:5d} K={K:5d}: n={n:3d} avg {us/n:8.1f} us  {fl/(us/n)/1e6:7.1f} TF   total {us:9.1f}")
print("total dense us", tot0)
