"""Phase stamps of the wave-specialised front kernel (probe build with -DWS_TRACE; s_memtime ticks of 10 ns)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from msclip_amd import hip, packing as P
BF = torch.bfloat16
B, S = 512, 224
g = torch.Generator().manual_seed(0)
img = torch.randn(B, 3, S, S, generator=g).cuda()
w, b = (torch.randn(27, 96, generator=g) * 0.3).cuda(), (torch.randn(96, generator=g) * 0.2).cuda()
Hm = S // 2
c2 = P.ConvSpec(torch.randn(96, 48, 3, 3, generator=g) * 0.07, torch.randn(96, generator=g) * 0.2, Hm, Hm, 2, 1).to("cuda")
ob = torch.empty(B * Hm * Hm, 48, dtype=BF, device="cuda")
out = torch.empty(B * c2.h_out ** 2, 96, dtype=BF, device="cuda")
for _ in range(3):
    hip.stem_dual_conv3x3s2(img, w, b, ob, c2.weight, c2.bias, out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
L = hip.lib()
L.msclip_front_trace_read.argtypes = [ctypes.c_void_p]
assert L.msclip_front_trace_read(ctypes.cast(buf, ctypes.c_void_p)) == 0
v = list(buf)
t0 = min(x for x in v[:14] if x)
names = {0: "P top", 1: "P parked", 2: "P after B1", 3: "P produced", 4: "P after B2",
         8: "C top", 9: "C half K", 10: "C after B1", 11: "C K done", 12: "C stored", 13: "C after B2"}
for k in sorted(names):
    print(f"{names[k]:14s} {(v[k] - t0) * 10:6d} ns")
