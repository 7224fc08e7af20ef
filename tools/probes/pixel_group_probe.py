"""Pointwise 48 -> 48 convolution over 6.4 M pixels (K padded 48 -> 64, 96-byte rows) against the same product with g pixels per
GEMM row and a block-diagonal filter (g * 48 columns: 192-byte / 384-byte rows, no K padding at g = 4)."""
import sys, time, torch
sys.path.insert(0, ".")
from msclip_amd import hip
BF = torch.bfloat16
M, C = 512 * 112 * 112, 48
g = torch.Generator().manual_seed(0)
x = torch.zeros(M * C + 256, dtype=BF, device="cuda")[:M * C].view(M, C)
x.copy_(torch.randn(M // 64, C, generator=g).to(BF).repeat(64, 1))
w = (torch.randn(C, C, generator=g) * 0.1).to(BF).cuda()
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e6
wp = torch.zeros(C, 64, dtype=BF, device="cuda"); wp[:, :C] = w
out = torch.zeros(M * C + 256, dtype=BF, device="cuda")[:M * C].view(M, C)
t1 = timed(lambda: hip.gemm(x, wp, out, M=M, N=C, ldx=C))
ref = out.clone()
print(f"plain [M, 48] x [48, 64]: {t1:.0f} us ({hip.gemm_variant(hip.describe_gemm(0, M, C, 64, 0, None, ldx=C))})")
for grp in (2, 4):
    K = grp * C
    Kp = (K + 63) // 64 * 64
    wb = torch.zeros(K, Kp, dtype=BF, device="cuda")
    for i in range(grp):
        wb[i * C:(i + 1) * C, i * C:(i + 1) * C] = w
    xg, og = x.view(M // grp, K), torch.zeros(M * C + 256, dtype=BF, device="cuda")[:M * C].view(M // grp, K)
    t = timed(lambda: hip.gemm(xg, wb, og, M=M // grp, N=K, ldx=K))
    print(f"{grp} pixels per row, K = {Kp}: {t:.0f} us, equal {torch.equal(og.view(M, C), ref)} ({hip.gemm_variant(hip.describe_gemm(0, M // grp, K, Kp, 0, None, ldx=K))})")
