"""msclip_gemm_splitk_tn on channel counts that are not whole 256-tiles (the conv side's narrow weight gradients)."""
import sys
import time
import torch
sys.path.insert(0, ".")
from msclip_amd import hip                                                 # noqa: E402
BF = torch.bfloat16
g = torch.Generator().manual_seed(5)
for T, No, Ni in [(50000, 96, 864), (50176, 48, 448), (8000, 192, 1728), (4097, 768, 96), (3000, 384, 192), (1605632, 96, 448), (401408, 96, 864)]:
    wide_dy = (torch.randn(T + 3, No + 16, generator=g) * 0.5).to(BF).cuda()
    wide_x = torch.randn(T + 3, Ni + 64, generator=g).to(BF).cuda()
    wide_dy[T:] = float("nan"); wide_x[T:] = float("nan")
    wide_dy[:, No:] = float("nan"); wide_x[:, Ni:] = float("nan")          # the columns right of the operands
    dy, x = wide_dy[:, :No], wide_x[:, :Ni]
    tiles = ((No + 255) // 256) * ((Ni + 255) // 256)
    S = max(1, min(256 // tiles, T // 2048))
    ref = dy[:T].float().t() @ x[:T].float()
    out = torch.full((No, Ni), float("nan"), device="cuda")
    hip.gemm_splitk_tn(dy, x, T, S, out=out)
    first = out.clone()
    hip.gemm_splitk_tn(dy, x, T, S, out=out)
    err = (out - ref).abs().max().item()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        hip.gemm_splitk_tn(dy, x, T, S, out=out)
    torch.cuda.synchronize(); t_tn = (time.perf_counter() - t0) / 5
    # the transposing path
    import msclip_amd.gradgemm as G
    import os
    os.environ["MSCLIP_WGRAD_TN"] = "0"
    dyc, xc = dy.contiguous(), x.contiguous()
    o2 = G.wgrad(dyc, xc, T)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        G.wgrad(dyc, xc, T)
    torch.cuda.synchronize(); t_tr = (time.perf_counter() - t0) / 5
    del os.environ["MSCLIP_WGRAD_TN"]
    print(f"T={T} {No}x{Ni} S={S}: max err {err:.3g} (ref absmax {ref.abs().max().item():.3g}) finite {bool(torch.isfinite(out).all())} "
          f"repeatable {torch.equal(out, first)}; TN {1e6 * t_tn:.0f} us, transposing path {1e6 * t_tr:.0f} us (err {(o2 - ref).abs().max().item():.3g})")
