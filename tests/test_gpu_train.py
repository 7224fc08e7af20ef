"""Backward pass on a real MI355X.

* every backward kernel against a plain fp32 PyTorch statement (autograd of the same op);
* the whole slice -- contrastive head, both projections, every transformer block of both towers with the modality-shared
  tensors' summed gradients, the token path of the lateral adapters, embeddings -- against autograd of the REAL reference
  (tests/golden/b32-yfcc-msclips.grads.npz, captured by tools/make_golden.py::grads_fixture on the golden batch);
* an AdamW step with the reference's parameter groups lowers the loss of the batch it was computed on.

Stated tolerance (bf16 GEMM operands and activations in BOTH passes, fp32 accumulation / statistics, against fp32
autograd of an fp32 forward): per gradient tensor, max error over the golden's 64-point sample <= 8 % of the tensor's
abs-max (median over the 211 tensors <= 3 %; measured 2.2 %, worst 5 %), abs-mean within 5 %, cosine >= 0.995 for the tensors stored in full.
LayerNorm BIAS gradients are column sums of nearly cancelling rows at the golden batch of 4 (abs-mean 15 x below the
matching weight gradient's): 25 % / 15 % / cosine 0.95 for those.  The error is dominated by the forward: the bf16 towers'
unit features differ from the fp32 reference's by ~1e-3, i.e. ~0.03 on a logit at T = 1/0.07, a few percent on every
softmax probability the gradient starts from.
Yardstick: the reference's OWN gradients move by about as much when it runs under torch.autocast(bfloat16) instead of
fp32 (tests/golden/ref_bf16_gradient_deviation.json, tools/ref_bf16_gradient_deviation.py: eval mode token side median
2.0 %, conv side median 5.2 % / worst 41 % / cosine >= 0.978; train mode 3.9 % and 11.7 % / 31 % / 0.959); the two
whole-chain tests also assert that this build is no further from the fp32 reference than that.
The convolutional side (stem, parallel branch, adapter convolutions, every BatchNorm's gamma / beta with FROZEN running
statistics -- the fixture is eval-mode autograd) is a third class: its gradients pass through up to ten ReLU masks evaluated
on bf16 activations and, at the golden batch of 4, its per-channel BatchNorm gradients are cancelling sums over few
pixels: sample error <= 25 % of the tensor's abs-max (measured, b32 / b16: median 4.7 / 3.7 %, worst 22 / 21 %), abs-mean
within 10 % (measured worst 3.7 / 6.9 %), cosine >= 0.975 (measured lowest 0.980 / 0.981).  The kernels underneath (im2col / col2im / depthwise gradients) are
pinned to 1e-5 against autograd of F.conv2d in the unit tests above."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden, set_opt, summarize, synth_sd, GOLDEN
from msclip_amd import hip, synth, train
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
SAMPLE_TOL, ABSMEAN_TOL, COS_TOL = 0.08, 0.05, 0.995
LNB_SAMPLE_TOL, LNB_ABSMEAN_TOL, LNB_COS_TOL = 0.25, 0.15, 0.95
CONV_SAMPLE_TOL, CONV_ABSMEAN_TOL, CONV_COS_TOL = 0.25, 0.10, 0.975
CONV_SIDE = ("resblocks.0.conv1", "resblocks.0.bn1", "resblocks.0.resnet_stage", "resblocks.0.last_conv", "parallel_branch_v",
             "top2bottom", "bottom_dw_conv")


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def rel(got, ref):
    return ((got.float() - ref.float()).abs().max() / ref.float().abs().max().clamp_min(1e-12)).item()


def close(got, ref, atol, rtol=0.0):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    assert bool((err <= atol + rtol * ref.abs()).all()), f"max err {err.max().item():.4g} (ref absmax {ref.abs().max().item():.4g})"


def test_transpose_cast_colsum_gelu(gpu_device):
    x = rnd(1000, 200, seed=1, dtype=BF)
    t = hip.transpose_bf16(x)
    assert t.shape == (200, 1024) and torch.equal(t[:, :1000], x.t()) and bool((t[:, 1000:] == 0).all())
    t2 = hip.transpose_bf16(x[100:900], 777)
    assert t2.shape == (200, 832) and torch.equal(t2[:, :777], x[100:877].t()) and bool((t2[:, 777:] == 0).all())
    x3 = rnd(130, 27, seed=5, dtype=BF)                               # odd width: the scalar tile kernel
    t3 = hip.transpose_bf16(x3)
    assert t3.shape == (27, 192) and torch.equal(t3[:, :130], x3.t()) and bool((t3[:, 130:] == 0).all())
    # many matrices in one launch (the training step's W^T operands); a row-range view as a source; repeated runs rewrite the outputs
    big = rnd(900, 3080, seed=6, dtype=BF)
    mats = [rnd(768, 3072, seed=7, dtype=BF), rnd(3072, 768, seed=8, dtype=BF), rnd(64, 8, seed=9, dtype=BF), big[64:832, 8:2312],
            rnd(2304, 776, seed=10, dtype=BF)[:, :768]]
    plan = hip.TransposePlan(mats)
    for rep in range(2):
        outs = plan.run()
        for m, o in zip(mats, outs):
            assert o.shape == (m.shape[1], m.shape[0]) and torch.equal(o, m.t())
        mats[0].mul_(2.0)
        mats[3].add_(1.0)
    f = rnd(300, 768, seed=2)
    assert torch.equal(hip.cast_bf16(f), f.to(BF))
    assert rel(hip.colsum(f), f.sum(0)) < 1e-5 and rel(hip.colsum(x), x.float().sum(0)) < 1e-5
    for mm, nn in ((5000, 48), (3000, 3072), (2500, 8), (700, 27), (4100, 104)):     # 16-byte kernel (narrow / wide / chunked) and scalar kernel
        xb = rnd(mm, nn, seed=6, dtype=BF)
        assert rel(hip.colsum(xb), xb.float().sum(0)) < 1e-5, (mm, nn)
        assert torch.equal(hip.colsum(xb), hip.colsum(xb))
    for mm, nn, take in ((70000 * 4, 48, None), (65536 + 32, 192, 65536), (66000, 384, 65600), (65536 * 2 + 2, 48, None)):
        xb = rnd(mm, nn, seed=9, dtype=BF)                              # narrow + long: r rows read as one wide row (with row slack)
        got = hip.colsum(xb, M=take)
        ref = xb[:take].double().sum(0).float() if take else xb.double().sum(0).float()
        assert rel(got, ref) < 2e-6, (mm, nn)
        acc2 = torch.ones(nn, device="cuda")
        hip.colsum(xb, out=acc2, M=take, accumulate=True)
        assert rel(acc2, ref + 1) < 2e-6
    part = rnd(7, 65536 + 64, seed=8)                                   # split-K partials: few rows, very wide -> the fold kernel
    assert rel(hip.colsum(part), part.double().sum(0).float()) < 1e-6 and torch.equal(hip.colsum(part), hip.colsum(part))
    acc = torch.ones(768, device="cuda")
    hip.colsum(f, out=acc, accumulate=True)
    assert rel(acc, f.sum(0) + 1) < 1e-5
    h, dy = rnd(64, 3072, seed=3, scale=2.0, dtype=BF), rnd(64, 3072, seed=4, dtype=BF)
    y, dh = torch.empty_like(h), torch.empty_like(h)
    hip.quickgelu(h, y)
    hf = h.float().requires_grad_(True)
    ref = hf * torch.sigmoid(1.702 * hf)
    assert rel(y, ref.detach()) < 1e-2
    ref.backward(dy.float())
    hip.quickgelu_bwd(h, dy, dh)
    assert rel(dh, hf.grad) < 1e-2


def test_colsum_single_launch_last_block_fold(gpu_device):
    """Chunked column sums are ONE launch since round 5: the workgroup that draws a column block's last ticket folds the partial
    rows in chunk order.  Against fp64 torch sums; bitwise repeatable over many launches (the arrival order of the workgroups
    changes, the result does not); the same from two streams at once (ticket counters come from a ring: no sharing); the
    accumulate form; and equal to the two-stage form's sum up to fp32 rounding."""
    for mm, nn, dt in ((43264, 2304, BF), (43264, 768, torch.float32), (17664, 3072, BF), (9000, 104, BF), (3000, 27, BF),
                       (2000, 1536, torch.float32), (800, 768, torch.float32)):
        x = rnd(mm, nn, seed=11, dtype=dt)
        ref = x.double().sum(0).float()
        first = hip.colsum(x)
        assert rel(first, ref) < 3e-6 * (8 if dt == BF else 1), (mm, nn, rel(first, ref))
        for _ in range(20):
            assert torch.equal(hip.colsum(x), first), (mm, nn)
        acc = torch.full((nn,), 2.0, device="cuda")
        hip.colsum(x, out=acc, accumulate=True)
        assert rel(acc, ref + 2) < 3e-6 * (8 if dt == BF else 1)
    # two streams: interleaved launches must not share ticket counters
    a, b = rnd(43264, 768, seed=12, dtype=BF), rnd(30000, 3072, seed=13, dtype=BF)
    ra, rb = hip.colsum(a), hip.colsum(b)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(10):
        with torch.cuda.stream(s1):
            outs.append((hip.colsum(a), ra))
        with torch.cuda.stream(s2):
            outs.append((hip.colsum(b), rb))
    torch.cuda.synchronize()
    assert all(torch.equal(g, r) for g, r in outs)


@pytest.mark.parametrize("M,C", [(65024, 768), (1000, 768), (37, 512), (5000, 1024)])
def test_cast_with_column_sums(gpu_device, M, C):
    """msclip_cast_bf16_colsum: the bf16 copy of msclip_cast_bf16 (bitwise) and the fp32 column sums of the same pass, on a
    row-range view of a wider buffer; rows past M untouched."""
    g = torch.Generator(device="cuda").manual_seed(3)
    buf = torch.randn(M + 5, C + 8, device="cuda", generator=g)
    x = buf[2:2 + M, :C]
    out = torch.full((M + 2, C), 7.0, dtype=torch.bfloat16, device="cuda")
    y, s = hip.cast_bf16_colsum(x, out[:M])
    assert torch.equal(y, hip.cast_bf16(x.contiguous())) and bool((out[M:] == 7.0).all())
    ref = x.double().sum(0)
    assert (s.double() - ref).abs().max().item() <= 2e-6 * x.abs().double().sum(0).max().item()
    assert torch.equal(s, hip.cast_bf16_colsum(x, out[:M])[1])                      # fixed order
    # skip_group: g + 1 rows per sample, the class row skipped (the lateral adapters' backward reads the grid rows of dsum)
    g = 7
    xs = torch.randn(9 * (g + 1), C, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    ys, ss = hip.cast_bf16_colsum(xs, skip_group=g)
    grid = xs.view(9, g + 1, C)[:, 1:].reshape(9 * g, C)
    assert ys.shape == (9 * g, C) and torch.equal(ys, grid.to(torch.bfloat16))
    assert (ss.double() - grid.double().sum(0)).abs().max().item() <= 2e-6 * grid.abs().double().sum(0).max().item()


def test_fold_plan_many_column_sums_in_one_launch(gpu_device):
    """hip.FoldPlan / msclip_colsum_multi (the training step's deferred folds of per-block partial matrices): > 96 items (two
    launches), row-range / column-range views, ragged widths, the scaled head of the packed in_proj bias; fixed summation order."""
    shapes = [(1024, 1536), (1024, 768), (320, 3072), (1024, 2304), (7, 5), (1, 64), (33, 100), (512, 2304)] * 13      # 104 items
    big = rnd(1100, 3100, seed=70)
    got = {}
    def build():
        plan = hip.FoldPlan(big.device)
        srcs = []
        for i, (M, N) in enumerate(shapes):
            src = big[i % 50:i % 50 + M, i % 20:i % 20 + N] if M <= 1024 and N <= 3072 else rnd(M, N, seed=i)
            kw = dict(scale_n=N // 3, scale=0.125) if i % 4 == 3 else {}
            plan.add(src, lambda r, i=i: got.__setitem__(i, r), **kw)
            srcs.append((src, kw))
        plan.run()
        return srcs
    srcs = build()
    first = {i: r.clone() for i, r in got.items()}
    for i, (src, kw) in enumerate(srcs):
        ref = src.double().sum(0)
        if kw:
            ref[:kw["scale_n"]] *= kw["scale"]
        assert got[i].shape == (src.shape[1],)
        assert (got[i].double() - ref).abs().max().item() <= 2e-6 * src.abs().double().sum(0).max().item() + 1e-12, i
    build()
    assert all(torch.equal(got[i], first[i]) for i in first)                    # bitwise repeatable


@pytest.mark.parametrize("C,dy_f32,gather", [(768, False, False), (768, True, False), (512, False, False), (768, False, True)])
def test_layernorm_backward(gpu_device, C, dy_f32, gather):
    M = 1000
    x = rnd(M * (3 if gather else 1), C, seed=5, scale=2.0) + 0.3
    gam, bet = rnd(C, seed=6) + 1.0, rnd(C, seed=7)
    dy = rnd(M, C, seed=8, dtype=torch.float32 if dy_f32 else BF)
    idx = (torch.arange(M, device="cuda") * 3 + 1).int() if gather else None
    xr = x[idx.long()] if gather else x
    xa = xr.clone().requires_grad_(True)
    ga, ba = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    u = xa.mean(-1, keepdim=True)
    s = (xa - u).pow(2).mean(-1, keepdim=True)
    (ga * ((xa - u) / torch.sqrt(s + 1e-12)) + ba).backward(dy.float())
    dx = torch.full_like(x, 0.5)
    dg, db = hip.layernorm_bwd(x, dy, gam, dx, M, row_idx=idx, accumulate=True)
    got = dx[idx.long()] if gather else dx
    assert rel(got - 0.5, xa.grad) < 2e-4
    if gather:
        mask = torch.ones(x.shape[0], dtype=torch.bool, device="cuda")
        mask[idx.long()] = False
        assert bool((dx[mask] == 0.5).all())                                  # untouched rows
    assert rel(dg, ga.grad) < 2e-4 and rel(db, ba.grad) < 2e-4
    dx2 = torch.full_like(x, 7.0)
    hip.layernorm_bwd(x, dy, gam, dx2, M, row_idx=idx, accumulate=False, want_param_grads=False)
    assert rel(dx2[idx.long()] if gather else dx2, xa.grad) < 2e-4
    if not gather:
        # the same pass with the bf16 copy of the written dx rows and their per-block column sums (training step: the next
        # projection's output gradient and bias gradient without a cast_bf16_colsum pass): dx, dgamma, dbeta bit for bit; two row
        # segments into one set of partials (second launch accumulates)
        dx3 = torch.full_like(x, 0.5)
        dxb = torch.full((M + 3, C), 7.0, dtype=BF, device="cuda")
        part = torch.full((hip.LN_PART_BLOCKS, C), float("nan"), dtype=torch.float32, device="cuda")
        cut = 336
        dg1, db1 = hip.layernorm_bwd(x[:cut], dy[:cut], gam, dx3[:cut], cut, dxb=dxb[:cut], sum_part=part)
        dg2, db2 = hip.layernorm_bwd(x[cut:], dy[cut:], gam, dx3[cut:], M - cut, dxb=dxb[cut:M], sum_part=part, sum_accumulate=True)
        assert torch.equal(dx3, dx) and bool((dxb[M:] == 7.0).all())
        assert torch.equal(dxb[:M], hip.cast_bf16(dx)) and bool(torch.isfinite(part).all())
        assert rel(dg1 + dg2, ga.grad) < 2e-4 and rel(db1 + db2, ba.grad) < 2e-4
        ref = dx.double().sum(0)
        assert (hip.colsum(part).double() - ref).abs().max().item() <= 2e-6 * dx.abs().double().sum(0).max().item()
        rc = hip.lib().msclip_layernorm_bwd(hip._p(x), x.stride(0), None, 1, hip._p(dy), dy.stride(0), int(dy_f32), hip._p(gam), hip._p(dx3),
                                            dx3.stride(0), 1, None, hip.LN_PART_BLOCKS, M, C, 1e-12, hip._p(dxb), C, None, 0, None)
        assert rc == -1                                                      # dxb without sum_part: refused


@pytest.mark.parametrize("L,causal", [(50, False), (77, True), (64, False), (33, True), (96, True), (1, False),
                                      (197, False), (197, True), (97, False), (160, True), (161, False), (208, True), (130, False)])
def test_attention_backward(gpu_device, L, causal):
    ns, H = 3, 12
    qkv = rnd(ns * L, 3 * H * 64, seed=9, scale=0.7, dtype=BF)
    dout = rnd(ns * L, H * 64, seed=10, dtype=BF)
    o = torch.empty(ns * L, H * 64, dtype=BF, device="cuda")
    hip.attention(qkv, o, ns, L, H, causal)
    qf = qkv.float().requires_grad_(True)
    q, k, v = (t.reshape(ns, L, H, 64).transpose(1, 2) for t in qf.chunk(3, dim=-1))
    sc = q @ k.transpose(-1, -2)                                              # q is pre-scaled in the packed layout
    if causal:
        sc = sc + torch.full((L, L), float("-inf"), device="cuda").triu_(1)
    ref_o = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(ns * L, H * 64)
    ref_o.backward(dout.float())
    dqkv = torch.full_like(qkv, float("nan"))
    hip.attention_bwd(qkv, o, dout, dqkv, ns, L, H, causal)
    assert bool(torch.isfinite(dqkv.float()).all())
    assert rel(dqkv, qf.grad) < 3e-2
    cos = F.cosine_similarity(dqkv.float().flatten(), qf.grad.flatten(), dim=0).item()
    assert cos > 0.999, cos
    if L <= 96:
        # the same launch with the per-sample token sums of dqkv (in_proj bias-gradient partials): dqkv bit for bit, the sums of the
        # fp32 values = the stored bf16 values' up to their rounding; many pairs per workgroup (persistent loop, two LDS buffers)
        ns2 = 67
        qkv2 = rnd(ns2 * L, 3 * H * 64, seed=19, scale=0.7, dtype=BF)
        dout2 = rnd(ns2 * L, H * 64, seed=20, dtype=BF)
        o2 = torch.empty(ns2 * L, H * 64, dtype=BF, device="cuda")
        hip.attention(qkv2, o2, ns2, L, H, causal)
        d_a, d_b = torch.full_like(qkv2, float("nan")), torch.full_like(qkv2, float("nan"))
        part = torch.full((ns2 + 1, 3 * H * 64), float("nan"), dtype=torch.float32, device="cuda")
        hip.attention_bwd(qkv2, o2, dout2, d_a, ns2, L, H, causal)
        hip.attention_bwd(qkv2, o2, dout2, d_b, ns2, L, H, causal, colsum_part=part[:ns2])
        assert torch.equal(d_a, d_b) and bool(torch.isnan(part[ns2:]).all())
        want = d_a.float().view(ns2, L, -1).sum(1)
        bound = d_a.float().abs().view(ns2, L, -1).sum(1) * 2.0 ** -8 + 1e-6
        assert bool(((part[:ns2] - want).abs() <= bound).all()), ((part[:ns2] - want).abs() / bound).max().item()
    else:
        with pytest.raises(hip.HipError):
            hip.attention_bwd(qkv, o, dout, dqkv, ns, L, H, causal, colsum_part=torch.empty(ns, 3 * H * 64, device="cuda"))


def _adapter_token_path_case(B, g, Cc, usecls):
    Lt = g * g + 1
    xin, tt = rnd(B * Lt, Cc, seed=17), rnd(B * g * g, Cc, seed=18)
    dww, dwb = rnd(9, Cc, seed=19, scale=0.3), rnd(Cc, seed=20)
    xa = xin.clone().requires_grad_(True)
    xv = xa.view(B, Lt, Cc)
    grid = xv[:, 1:].transpose(1, 2).reshape(B, Cc, g, g)
    bo = F.conv2d(grid, dww.t().reshape(Cc, 1, 3, 3), dwb, padding=1, groups=Cc).flatten(2).transpose(1, 2)
    ref_sum = torch.cat([(2 if usecls else 1) * xv[:, :1], bo + tt.view(B, g * g, Cc)], 1).reshape(B * Lt, Cc)
    out = torch.empty_like(xin)
    hip.adapter_sum(xin, tt, dww, dwb, out, B, Lt, g, usecls)
    assert rel(out, ref_sum.detach()) < 1e-5, (B, g, Cc)
    dsum = rnd(B * Lt, Cc, seed=21)
    ref_sum.backward(dsum)
    dxa = torch.empty_like(xin)
    hip.adapter_dx(dsum, dww, dxa, B, Lt, g, usecls)
    assert rel(dxa, xa.grad) < 1e-5, (B, g, Cc)


def test_l2norm_loss_embed_adapter_adamw(gpu_device):
    # l2norm
    x, dy = rnd(37, 512, seed=11), rnd(37, 512, seed=12)
    xa = x.clone().requires_grad_(True)
    (xa / xa.norm(dim=-1, keepdim=True)).backward(dy)
    dx = torch.empty_like(x)
    hip.l2norm_bwd(x, dy, dx)
    assert rel(dx, xa.grad) < 1e-4
    # dL/dS of the symmetric CE on a row block with a label offset
    R, N, off = 40, 100, 30
    S = rnd(R, N, seed=13, scale=3.0)
    full = rnd(N, N, seed=14, scale=3.0)
    full[off:off + R] = S
    fa = full.clone().requires_grad_(True)
    lab = torch.arange(N, device="cuda")
    (0.5 * (F.cross_entropy(fa, lab) + F.cross_entropy(fa.t(), lab))).backward()
    G = torch.empty(R, 128, dtype=BF, device="cuda")
    dsp = torch.empty(R, device="cuda")
    hip.clip_loss_bwd_g(S, torch.logsumexp(full, 1)[off:off + R].contiguous(), torch.logsumexp(full, 0).contiguous(), off,
                        1.0 / (2 * N), G, dsp)
    assert rel(G[:, :N], fa.grad[off:off + R]) < 1e-2 and bool((G[:, N:] == 0).all())
    assert abs(dsp.sum().item() - (fa.grad[off:off + R] * S).sum().item()) < 1e-3
    # embedding backward
    tok = torch.randint(0, 50, (6, 77), generator=torch.Generator().manual_seed(15)).cuda()
    dxe = rnd(6 * 77, 768, seed=16)
    demb, dpos = torch.zeros(50, 768, device="cuda"), torch.zeros(77, 768, device="cuda")
    hip.embed_tokens_bwd(tok, dxe, demb, dpos)
    ref = torch.zeros(50, 768, device="cuda").index_add_(0, tok.flatten(), dxe)
    assert rel(demb, ref) < 1e-5 and rel(dpos, dxe.view(6, 77, 768).sum(0)) < 1e-5
    # lateral adapter token path (M.py:1763-1777): sum and its input gradient against conv2d autograd
    for B, g, Cc, usecls in ((3, 7, 768, True), (2, 14, 1100, False), (2, 5, 50, True)):   # 16-byte form, > 1024 channels, 4-byte form
        _adapter_token_path_case(B, g, Cc, usecls)
    B, g, Cc = 3, 7, 768
    Lt = g * g + 1
    xin, tt = rnd(B * Lt, Cc, seed=17), rnd(B * g * g, Cc, seed=18)
    dww, dwb = rnd(9, Cc, seed=19, scale=0.3), rnd(Cc, seed=20)
    xa = xin.clone().requires_grad_(True)
    xv = xa.view(B, Lt, Cc)
    grid = xv[:, 1:].transpose(1, 2).reshape(B, Cc, g, g)
    bo = F.conv2d(grid, dww.t().reshape(Cc, 1, 3, 3), dwb, padding=1, groups=Cc).flatten(2).transpose(1, 2)
    ref_sum = torch.cat([2 * xv[:, :1], bo + tt.view(B, g * g, Cc)], 1).reshape(B * Lt, Cc)
    out = torch.empty_like(xin)
    hip.adapter_sum(xin, tt, dww, dwb, out, B, Lt, g, True)
    assert rel(out, ref_sum.detach()) < 1e-5
    dsum = rnd(B * Lt, Cc, seed=21)
    ref_sum.backward(dsum)
    dxa = torch.empty_like(xin)
    hip.adapter_dx(dsum, dww, dxa, B, Lt, g, True)
    assert rel(dxa, xa.grad) < 1e-5
    # AdamW against torch.optim.AdamW
    p0, gr = rnd(5000, seed=22), rnd(5000, seed=23)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for step in (1, 2, 3):
        pt.grad = gr * step
        opt.step()
        hip.adamw(p, gr * step, m, v, 1e-3, 0.9, 0.98, 1e-6, 0.2, step)
    assert rel(p, pt.detach()) < 1e-5


def test_adamw_multi_equals_per_tensor_calls(gpu_device):
    """msclip_adamw_multi over a ragged list (1 element ... 20 M: tensors that span several launches, more tensors than one
    launch's table, an unaligned gradient view as the bucketed all-reduce hands them out) = msclip_adamw tensor by tensor,
    bit for bit, over three steps; guard elements behind every tensor stay untouched."""
    sizes = [1, 3, 768, 5000, 32768, 32769, 65536 + 5, 768 * 768, 20_000_003] + [100 + 7 * i for i in range(60)]
    gen = torch.Generator(device="cuda").manual_seed(5)
    flat = torch.randn(sum(sizes) + len(sizes) + 1, device="cuda", generator=gen)          # gradients as views at odd offsets
    ps, offs, o = [], [], 1
    for n in sizes:
        ps.append(torch.randn(n + 4, device="cuda", generator=gen))
        offs.append(o)
        o += n + 1
    one = [(p.clone(), torch.zeros_like(p), torch.zeros_like(p)) for p in ps]
    many = [(p.clone(), torch.zeros_like(p), torch.zeros_like(p)) for p in ps]
    for step in (1, 2, 3):
        fs = flat * step
        gs = [fs[o:o + n] for o, n in zip(offs, sizes)]
        assert any(g.data_ptr() % 16 for g in gs)
        for (p, m, v), g, n in zip(one, gs, sizes):
            hip.adamw(p[:n], g, m[:n], v[:n], 1e-3 * (1 + n % 3), 0.9, 0.98, 1e-6, 0.2 * (n % 2), step)
        hip.adamw_multi([(p[:n], g, m[:n], v[:n], 1e-3 * (1 + n % 3), 0.2 * (n % 2))
                         for (p, m, v), g, n in zip(many, gs, sizes)], 0.9, 0.98, 1e-6, step)
    for (p1, m1, v1), (p2, m2, v2), p0, n in zip(one, many, ps, sizes):
        assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2), (n, (p1 - p2).abs().max().item(), (m1 - m2).abs().max().item())
        assert torch.equal(p2[n:], p0[n:]) and not torch.equal(p2[:n], p0[:n])
    with pytest.raises(hip.HipError):
        hip.adamw_multi([(one[0][0], one[0][0], one[0][1], one[0][2], 1e-3, 0.0)], 0.9, 0.98, 1e-6, 0)


def test_adamw_multi_packed_copies(gpu_device):
    """msclip_adamw_tensor.pk: the kernel also writes p_new * pk_scale as bf16 (the engine's projection operands, q rows
    carrying 64^-0.5) or fp32 (the scaled in_proj bias): bitwise what a cast of the updated tensor gives; tensors that span
    launches and unaligned gradient views included; the parameter update itself is unchanged."""
    sizes = [8, 768 * 768, 3 * 32768 + 12, 20_000_003, 5000, 7]
    gen = torch.Generator(device="cuda").manual_seed(9)
    flat = torch.randn(sum(sizes) + len(sizes) + 1, device="cuda", generator=gen)
    offs, o = [], 1
    for n in sizes:
        offs.append(o)
        o += n + 1
    gs = [flat[o:o + n] for o, n in zip(offs, sizes)]
    ps = [torch.randn(n, device="cuda", generator=gen) for n in sizes]
    plain = [(p.clone(), torch.zeros_like(p), torch.zeros_like(p)) for p in ps]
    packd = [(p.clone(), torch.zeros_like(p), torch.zeros_like(p)) for p in ps]
    kinds = [(torch.bfloat16, 0.125), (torch.bfloat16, 1.0), (torch.float32, 0.125), (torch.bfloat16, 0.125), (None, 1.0), (torch.float32, 1.0)]
    pks = [None if dt is None else torch.full((n + 8,), 7.0, dtype=dt, device="cuda") for (dt, _), n in zip(kinds, sizes)]
    plan = hip.AdamwPlan([(p, g, m, v, 1e-3, 0.1, None if pk is None else pk[:n], sc)
                          for (p, m, v), g, pk, (_, sc), n in zip(packd, gs, pks, kinds, sizes)])
    for step in (1, 2):
        hip.adamw_multi([(p, g, m, v, 1e-3, 0.1) for (p, m, v), g in zip(plain, gs)], 0.9, 0.999, 1e-8, step)
        plan.run(0.9, 0.999, 1e-8, step)
    for (p1, _, _), (p2, _, _), pk, (dt, sc), n in zip(plain, packd, pks, kinds, sizes):
        assert torch.equal(p1, p2)
        if pk is not None:
            assert torch.equal(pk[:n], (p2 * sc).to(dt)), (n, dt)
            assert bool((pk[n:] == 7.0).all())


@pytest.mark.parametrize("B,S,dt", [(3, 32, torch.float32), (2, 224, torch.float32), (5, 38, BF)])
def test_stem_dual_conv_raw_outputs(gpu_device, B, S, dt):
    """msclip_stem_conv3x3s2_dual_raw (train-mode BatchNorm's forward of the two convolutions on the image): RAW fp32 outputs of both
    from one pass -- against F.conv2d on the bf16-rounded image and filters, and against the patch-matrix GEMM it replaces."""
    Ho = S // 2
    img = rnd(B, 3, S, S, seed=51).to(dt)
    wa, wb = rnd(48, 3, 3, 3, seed=52, scale=0.3), rnd(48, 3, 3, 3, seed=53, scale=0.3)
    w = torch.cat([wa.reshape(48, 27), wb.reshape(48, 27)], 0).t().contiguous()
    oa = torch.full((B * Ho * Ho + 3, 48), 7.0, device="cuda")
    ob = torch.full((B * Ho * Ho + 3, 48), 7.0, device="cuda")
    hip.stem_conv_dual_raw(img, w, oa, ob)
    xr = img.to(BF).float()
    for o, wt in ((oa, wa), (ob, wb)):
        ref = F.conv2d(xr, wt.to(BF).float(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(B * Ho * Ho, 48)
        assert rel(o[:B * Ho * Ho], ref) < 2e-5, rel(o[:B * Ho * Ho], ref)
        assert bool((o[B * Ho * Ho:] == 7.0).all())
    col = hip.im2col(img, B, S, S, 3, 3, 3, 2, 1, image=True)
    wm = torch.zeros(48, 64, dtype=BF, device="cuda")
    wm[:, :27] = wa.permute(0, 2, 3, 1).reshape(48, 27).to(BF)
    r = torch.empty(B * Ho * Ho, 48, device="cuda")
    hip.gemm(col, wm, r)
    assert rel(oa[:B * Ho * Ho], r) < 2e-5


@pytest.mark.parametrize("B,S,dt", [(3, 32, torch.float32), (2, 224, torch.float32), (5, 38, BF), (16, 224, torch.float32)])
def test_stem_dual_conv_two_pass_batchnorm(gpu_device, B, S, dt):
    """msclip_stem_conv3x3s2_dual_stats / _norm (round 6): train-mode BatchNorm of both image convolutions without a raw map --
    against the raw-map path (msclip_stem_conv3x3s2_dual_raw + msclip_bn_stats + msclip_bn_apply) and against F.batch_norm on
    F.conv2d of the bf16-rounded operands; the normalised values the backward reads; nothing written behind the maps."""
    Ho = S // 2
    pix = B * Ho * Ho
    img = rnd(B, 3, S, S, seed=61).to(dt)
    wa, wb = rnd(48, 3, 3, 3, seed=62, scale=0.3), rnd(48, 3, 3, 3, seed=63, scale=0.3)
    w = torch.cat([wa.reshape(48, 27), wb.reshape(48, 27)], 0).t().contiguous()
    aff = [(rnd(48, seed=64 + k) * 0.5 + 1.0, rnd(48, seed=66 + k)) for k in range(2)]
    ys = [torch.full((pix + 3, 48), 7.0, dtype=BF, device="cuda") for _ in range(2)]
    xh = [torch.full((pix + 3, 48), 7.0, dtype=BF, device="cuda") for _ in range(2)]
    res = hip.stem_conv_dual_bn(img, w, aff, ys[0], ys[1], xh[0], xh[1], eps=1e-5)
    raws = [torch.empty(pix, 48, device="cuda") for _ in range(2)]
    hip.stem_conv_dual_raw(img, w, raws[0], raws[1])
    xr = img.to(BF).float()
    for k, wt in enumerate((wa, wb)):
        gam, bet = aff[k]
        mean, var, rstd, scale, shift = hip.bn_stats(raws[k], gamma=gam, beta=bet, eps=1e-5)
        assert rel(res[k][0], mean) <= 2e-5 and rel(res[k][1], var) <= 1e-4 and rel(res[k][3], scale) <= 1e-4
        want = torch.empty(pix, 48, dtype=BF, device="cuda")
        hip.bn_apply(raws[k], scale, shift, want, relu=True)
        # the same expression on the same fp32 accumulators with statistics equal to 1e-5: bf16 values equal but for rounding ties
        assert rel(ys[k][:pix], want) <= 8e-3 and (ys[k][:pix].float() != want.float()).float().mean().item() <= 2e-3
        assert rel(xh[k][:pix], (raws[k] - mean) * rstd) <= 6e-3
        ref = F.batch_norm(F.conv2d(xr, wt.to(BF).float(), stride=2, padding=1), None, None, gam, bet, training=True, eps=1e-5)
        close(ys[k][:pix], torch.relu(ref).permute(0, 2, 3, 1).reshape(pix, 48), 2e-2, 1e-2)
        assert bool((ys[k][pix:] == 7.0).all()) and bool((xh[k][pix:] == 7.0).all())
        # the backward from xhat (bf16; mean 0, rstd 1, gamma := scale) against the backward from the raw fp32 map
        dy = rnd(pix, 48, seed=70 + k, dtype=BF)
        dx_raw, dx_hat = (torch.empty(pix, 48, dtype=BF, device="cuda") for _ in range(2))
        (dg0, db0), = hip.bn_bwd_fused(dy, [(raws[k], mean, rstd, gam, dx_raw)], y=ys[k][:pix])
        zero, one = torch.zeros(48, device="cuda"), torch.ones(48, device="cuda")
        (dg1, db1), = hip.bn_bwd_fused(dy, [(xh[k][:pix].contiguous(), zero, one, scale.contiguous(), dx_hat)], y=ys[k][:pix])
        # (dgamma = sum d xhat is a cancelling sum for a random d: xhat's bf16 rounding, 2^-9 per term, shows undiminished)
        assert rel(dg1, dg0) <= 8e-3 and rel(db1, db0) <= 1e-5 and rel(dx_hat, dx_raw) <= 1e-2, (rel(dg1, dg0), rel(db1, db0), rel(dx_hat, dx_raw))


@pytest.mark.parametrize("ci,co,k,stride,H,B,resid", [(48, 48, 1, 1, 56, 3, False), (48, 96, 1, 2, 56, 6, True), (48, 96, 3, 2, 56, 6, False),
                                                      (48, 48, 3, 2, 40, 11, False), (96, 192, 1, 2, 28, 22, True), (96, 96, 1, 1, 28, 7, False),
                                                      (192, 384, 1, 1, 14, 24, True)])
def test_conv_two_pass_batchnorm_on_the_streaming_kernel(gpu_device, ci, co, k, stride, H, B, resid):
    """msclip_gemm_desc.bn_mode 1 / 2 (round 6): train-mode BatchNorm of a convolution the streaming kernel runs, without a raw map --
    column sums from a first pass (per-wave partial rows), normalise + optional residual + ReLU and xhat from a second -- against the
    raw fp32 map + msclip_bn_stats + msclip_bn_apply path on the same operands."""
    from msclip_amd import packing as P
    pad = 1 if k == 3 else 0
    x = rnd(B * H * H + 8, ci, seed=81, dtype=BF)                      # NHWC rows (+ slack for the K padding of pointwise launches)
    wt = rnd(co, ci, k, k, seed=82, scale=(2.0 / (ci * k * k)) ** 0.5)
    spec = P.ConvSpec(wt, torch.zeros(co), H, H, stride, pad).to("cuda")
    M = B * spec.h_out * spec.w_out
    pointwise = k == 1 and stride == 1
    kw = dict(M=M, N=co, ldx=ci) if pointwise else dict(M=M, N=co, conv=spec.geometry(), ktab=spec.ktab)
    assert hip.gemm_bn_two_pass_ok(ci, co, spec.weight.shape[1], M, conv=None if pointwise else spec.geometry())
    gam, bet = rnd(co, seed=83) * 0.5 + 1.0, rnd(co, seed=84)
    res = rnd(M, co, seed=85, dtype=BF) if resid else None
    raw = torch.empty(M, co, device="cuda")
    hip.gemm(x, spec.weight, raw, **kw)
    mean, var, rstd, scale, shift = hip.bn_stats(raw, gamma=gam, beta=bet, eps=1e-6)
    want = torch.empty(M, co, dtype=BF, device="cuda")
    hip.bn_apply(raw, scale, shift, want, relu=True, resid=res)
    part = torch.zeros(2048, 2 * co, device="cuda")
    hip.gemm(x, spec.weight, part, bn_stats_part=part, **kw)
    o = torch.empty(5, co, device="cuda")
    hip.bn_finish(hip.colsum(part), co, M, gam, bet, 1e-6, o)
    assert rel(o[0], mean) <= 2e-5 and rel(o[1], var) <= 1e-4 and rel(o[3], scale) <= 1e-4 and rel(o[4], shift) <= 1e-4
    y = torch.full((M + 2, co), 7.0, dtype=BF, device="cuda")
    xh = torch.full((M + 2, co), 7.0, dtype=BF, device="cuda")
    hip.gemm(x, spec.weight, y[:M], out2=xh[:M], bn_consts=o, act=hip.ACT_RELU, resid=res,
             resid_kind=hip.RESID_BF16 if resid else hip.RESID_NONE, **kw)
    assert rel(y[:M], want) <= 8e-3 and (y[:M].float() != want.float()).float().mean().item() <= 2e-3
    assert rel(xh[:M], (raw - mean) * rstd) <= 6e-3
    assert bool((y[M:] == 7.0).all()) and bool((xh[M:] == 7.0).all())


@pytest.mark.parametrize("B,S,co", [(3, 32, 48), (2, 224, 24), (5, 36, 16), (2, 64, 64), (1, 256, 40)])
def test_image_conv_wgrad_without_a_patch_matrix(gpu_device, B, S, co):
    """msclip_image_conv_wgrad (the stem's conv1 / parallel stage 0 on the input image): dW and the bias sums from one pass over dy
    and the image, against autograd of F.conv2d on the bf16-rounded image and against the patch-matrix path; odd output sides,
    channel counts below a 16-tile, the image borders."""
    Ho = (S - 1) // 2 + 1
    img = rnd(B, 3, S, S, seed=41)
    dy = rnd(B * Ho * Ho + 5, co, seed=42, dtype=BF)[:B * Ho * Ho]
    assert hip.image_conv_wgrad_ok(img, dy)
    dw, db = hip.image_conv_wgrad(img, dy)
    assert dw.shape == (co, 27) and db.shape == (co,)
    xr = img.to(BF).float()
    wr = torch.zeros(co, 3, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(xr, wr, stride=2, padding=1).backward(dy.float().view(B, Ho, Ho, co).permute(0, 3, 1, 2))
    ref = wr.grad.permute(0, 2, 3, 1).reshape(co, 27)                          # [co, ci, kh, kw] -> column (kh * 3 + kw) * 3 + ci
    assert rel(dw, ref) < 2e-5, rel(dw, ref)
    assert rel(db, dy.float().sum(0)) < 2e-5
    col = hip.im2col(img, B, S, S, 3, 3, 3, 2, 1, image=True)
    assert rel(dw, dy.float().t() @ col.float()[:, :27]) < 2e-5
    dw2, db2 = hip.image_conv_wgrad(img, dy)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                     # fixed summation order
    assert not hip.image_conv_wgrad_ok(img.to(BF), dy) and not hip.image_conv_wgrad_ok(img, dy[:, :co - 4])


@pytest.mark.parametrize("geom", [(2, 12, 16, 3, 2, 1), (3, 9, 8, 1, 2, 0), (2, 8, 24, 3, 1, 1), (2, 10, 3, 3, 2, 1), (3, 32, 3, 3, 2, 1),
                                  (2, 14, 3, 5, 1, 2)])
def test_im2col_col2im_against_conv_autograd(gpu_device, geom):
    """dW = dY^T . im2col(X) and dX = col2im(dY . W) against autograd of F.conv2d (fp32 on the same bf16 values);
    the last geometry is the 3-channel NCHW image path (no input gradient)."""
    B, H, C, k, s, pad = geom
    co = 16
    image = C == 3
    Ho = (H + 2 * pad - k) // s + 1
    x = rnd(B, C, H, H, seed=1).to(BF).float()                         # NCHW values exactly representable in bf16
    wgt = rnd(co, C, k, k, seed=2, scale=0.3).to(BF).float()
    dy = rnd(B, co, Ho, Ho, seed=3).to(BF).float()
    xr = x.clone().requires_grad_(True)
    wr = wgt.clone().requires_grad_(True)
    F.conv2d(xr, wr, stride=s, padding=pad).backward(dy)
    if image:
        col = hip.im2col(x.contiguous(), B, H, H, C, k, k, s, pad, image=True)
        # the bf16 image and the narrower padding (what the token-major wgrad GEMM reads) hold the same columns
        assert torch.equal(hip.im2col(x.to(BF).contiguous(), B, H, H, C, k, k, s, pad, image=True), col)
        narrow = hip.im2col(x.contiguous(), B, H, H, C, k, k, s, pad, image=True, kalign=8)
        assert narrow.shape[1] == (k * k * C + 7) // 8 * 8 and torch.equal(narrow, col[:, :narrow.shape[1]])
    else:
        col = hip.im2col(x.permute(0, 2, 3, 1).contiguous().to(BF).view(B * H * H, C), B, H, H, C, k, k, s, pad)
    K = k * k * C
    assert col.shape == (B * Ho * Ho, (K + 63) // 64 * 64) and torch.all(col[:, K:] == 0)
    dy2 = dy.permute(0, 2, 3, 1).reshape(B * Ho * Ho, co)
    dw = (dy2.t() @ col.float()[:, :K]).view(co, k, k, C).permute(0, 3, 1, 2)
    assert rel(dw, wr.grad) <= 1e-5
    if not image:
        wm = wgt.permute(0, 2, 3, 1).reshape(co, K)                     # packed forward layout
        dcol = torch.zeros(B * Ho * Ho, col.shape[1], dtype=BF, device="cuda")
        dcol[:, :K] = (dy2 @ wm).to(BF)
        dx = torch.full((B * H * H, C), 7.0, dtype=BF, device="cuda")
        hip.col2im(dcol, dx, B, H, H, C, k, k, s, pad)
        ref = xr.grad.permute(0, 2, 3, 1).reshape(B * H * H, C)
        assert rel(dx, ref) <= 2e-2                                     # dcol is rounded to bf16 before the tap sum
        dx2 = dx.clone()
        hip.col2im(dcol, dx2, B, H, H, C, k, k, s, pad, accumulate=True)
        assert rel(dx2, 2 * ref) <= 2e-2


@pytest.mark.parametrize("shape", [(48, 64, 8192, 8), (96, 448, 4096, 4), (200, 130, 2048, 1), (768, 48, 1024, 16)])
def test_gemm_splitk(gpu_device, shape):
    """msclip_gemm_splitk: K slices contracted by separate workgroup rows of one launch, folded in a fixed order."""
    M, N, K, S = shape
    a, b = rnd(M, K, seed=1).to(BF), rnd(N, K, seed=2).to(BF)
    got = hip.gemm_splitk(a, b, S)
    ref = a.double() @ b.double().t()
    assert rel(got, ref) <= 2e-6
    assert torch.equal(got, hip.gemm_splitk(a, b, S))                  # deterministic


def test_relu_dwpool_dw3x3_backward_kernels(gpu_device):
    B, g, k, C, D = 3, 7, 4, 48, 64
    H = g * k
    # relu mask with a second addend
    y = rnd(B * 49, C, seed=1).to(BF)
    d1, d2 = rnd(B * 49, C, seed=2).to(BF), rnd(B * 49, C, seed=3).to(BF)
    got = hip.relu_bwd(d1, y, dy2=d2)
    assert rel(got, ((d1.float() + d2.float()) * (y.float() > 0)).to(BF)) == 0
    assert torch.equal(hip.relu_bwd(d1, y), (d1.float() * (y.float() > 0)).to(BF))
    # depthwise kernel == stride conv
    top = rnd(B, C, H, H, seed=4).to(BF).float()
    wd = rnd(C, 1, k, k, seed=5, scale=0.2)
    dpool = rnd(B, C, g, g, seed=6).to(BF).float()
    tr, wr = top.clone().requires_grad_(True), wd.clone().requires_grad_(True)
    F.conv2d(tr, wr, stride=k, groups=C).backward(dpool)
    top_n = top.permute(0, 2, 3, 1).contiguous().to(BF).view(B * H * H, C)
    dp_n = dpool.permute(0, 2, 3, 1).contiguous().to(BF).view(B * g * g, C)
    wtab = wd[:, 0].reshape(C, k * k).t().contiguous()                  # [k*k, C] like packing.adapter_weights
    dw = hip.dwpool_wgrad(dp_n, top_n, B, H, H, C, k)
    assert rel(dw.t().reshape(C, 1, k, k), wr.grad) <= 1e-5
    dtop = torch.full((B * H * H, C), 3.0, dtype=BF, device="cuda")
    hip.dwpool_bwd(dp_n, wtab, dtop, B, H, H, C, k)
    ref = tr.grad.permute(0, 2, 3, 1).reshape(B * H * H, C)
    assert rel(dtop, ref) <= 8e-3
    hip.dwpool_bwd(dp_n, wtab, dtop, B, H, H, C, k, accumulate=True)
    assert rel(dtop, 2 * ref) <= 1.2e-2
    # depthwise 3x3 over the token grid (cls row excluded)
    L = g * g + 1
    x = rnd(B * L, D, seed=7)
    dsum = rnd(B * L, D, seed=8)
    wb = rnd(D, 1, 3, 3, seed=9).requires_grad_(True)
    grid = x.view(B, L, D)[:, 1:].transpose(1, 2).reshape(B, D, g, g)
    out = F.conv2d(grid, wb, padding=1, groups=D)
    out.backward(dsum.view(B, L, D)[:, 1:].transpose(1, 2).reshape(B, D, g, g))
    got = hip.dw3x3_wgrad(dsum, x, B, L, g)
    assert rel(got.t().reshape(D, 1, 3, 3), wb.grad) <= 1e-5


@pytest.mark.parametrize("B,g,k,C", [(5, 7, 16, 48), (4, 7, 2, 384), (6, 7, 1, 768), (2, 14, 8, 48), (3, 5, 3, 200)])
def test_dwpool_wgrad_geometries(gpu_device, B, g, k, C):
    """msclip_dwpool_wgrad over the adapters' real geometries (k = 16 ... 1, 48 ... 768 channels: several positions side by
    side per block below 129 channels, several channel passes above 256) against autograd of the strided depthwise conv."""
    H = g * k
    top = rnd(B, C, H, H, seed=4).to(BF).float()
    wd = rnd(C, 1, k, k, seed=5, scale=0.2).requires_grad_(True)
    dpool = rnd(B, C, g, g, seed=6).to(BF).float()
    F.conv2d(top, wd, stride=k, groups=C).backward(dpool)
    top_n = top.permute(0, 2, 3, 1).contiguous().to(BF).view(B * H * H, C)
    dp_n = dpool.permute(0, 2, 3, 1).contiguous().to(BF).view(B * g * g, C)
    dw = hip.dwpool_wgrad(dp_n, top_n, B, H, H, C, k)
    assert rel(dw.t().reshape(C, 1, k, k), wd.grad) <= 1e-5
    assert torch.equal(dw, hip.dwpool_wgrad(dp_n, top_n, B, H, H, C, k))          # deterministic


@pytest.mark.parametrize("B,g,D", [(5, 7, 768), (300, 7, 64), (3, 14, 768), (4, 5, 100), (2, 14, 300)])
def test_dw3x3_wgrad_geometries(gpu_device, B, g, D):
    """msclip_dw3x3_wgrad: the row form of the 7 x 7 and 14 x 14 token grids (all nine taps per thread) and the generic
    form, more samples than slabs, channel counts that are no multiple of the block."""
    L = g * g + 1
    x = rnd(B * L, D, seed=7)
    dsum = rnd(B * L, D, seed=8)
    wb = rnd(D, 1, 3, 3, seed=9).requires_grad_(True)
    grid = x.view(B, L, D)[:, 1:].transpose(1, 2).reshape(B, D, g, g)
    F.conv2d(grid, wb, padding=1, groups=D).backward(dsum.view(B, L, D)[:, 1:].transpose(1, 2).reshape(B, D, g, g))
    got = hip.dw3x3_wgrad(dsum, x, B, L, g)
    assert rel(got.t().reshape(D, 1, 3, 3), wb.grad) <= 2e-5
    assert torch.equal(got, hip.dw3x3_wgrad(dsum, x, B, L, g))


@pytest.mark.parametrize("dtype,M,C", [(BF, 5000, 48), (torch.float32, 3000, 768), (BF, 300, 96)])
def test_batchnorm_train_kernels(gpu_device, dtype, M, C):
    """msclip_bn_stats / _apply / _bwd_reduce / _bwd_dx against autograd of F.batch_norm(training=True)."""
    x = (rnd(M, C, seed=1) * 1.5 + 0.3).to(dtype)
    gam, bet = rnd(C, seed=2) * 0.5 + 1.0, rnd(C, seed=3)
    dy = rnd(M, C, seed=4).to(dtype)
    res = rnd(M, C, seed=5, dtype=BF)
    xr, gr, br = x.float().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    ref = F.batch_norm(xr, None, None, gr, br, training=True, eps=1e-5)
    ref.backward(dy.float())
    mean, var = hip.bn_stats(x)
    assert rel(mean, x.float().mean(0)) <= 1e-5 and rel(var, x.float().var(0, unbiased=False)) <= 1e-4
    rstd = torch.rsqrt(var + 1e-5)
    scale, shift = (gam * rstd).contiguous(), (bet - mean * gam * rstd).contiguous()
    out = torch.empty(M, C, dtype=BF, device="cuda")
    hip.bn_apply(x, scale, shift, out)
    assert rel(out, ref.detach().to(BF)) <= 8e-3
    out2 = torch.empty(M, C, dtype=torch.float32, device="cuda")
    hip.bn_apply(x, scale, shift, out2, relu=True, resid=res)
    assert rel(out2, torch.relu(ref.detach() + res.float())) <= 1e-4
    dx = torch.empty_like(x)
    dg, db = hip.bn_bwd(dy, x, mean.contiguous(), rstd.contiguous(), gam, dx)
    assert rel(dg, gr.grad) <= 1e-4 and rel(db, br.grad) <= 1e-4
    assert rel(dx, xr.grad) <= (1e-4 if dtype == torch.float32 else 1e-2)
    if dtype == torch.float32:                                        # fp32 raw conv output with a bf16 gradient (the stem / branch case)
        dyb = dy.to(BF)
        dxb = torch.empty_like(dyb)
        hip.bn_bwd(dyb, x, mean.contiguous(), rstd.contiguous(), gam, dxb)
        xr2 = x.detach().clone().requires_grad_(True)
        F.batch_norm(xr2, None, None, gam, bet, training=True, eps=1e-5).backward(dyb.float())
        assert rel(dxb, xr2.grad) <= 1e-2


@pytest.mark.parametrize("M,C,two,mask,add", [(6272, 48, True, True, True), (3000, 768, False, True, False), (4096, 96, True, True, False),
                                              (777, 192, False, False, True), (50176, 48, False, True, True)])
def test_batchnorm_backward_fused_with_relu_mask(gpu_device, M, C, two, mask, add):
    """msclip_bn_bwd_fused (round 6): d = bf16(dy [+ dy2]) * (y > 0) formed on the fly, one or two BatchNorms per pass, four columns
    per thread -- against msclip_relu_bwd followed by one msclip_bn_bwd_reduce / _dx pair per BatchNorm (the round-5 path, itself
    pinned to autograd of F.batch_norm above), row-folded narrow maps and a ragged row count included."""
    dy, dy2 = rnd(M, C, seed=1, dtype=BF), rnd(M, C, seed=2, dtype=BF)
    y = torch.relu(rnd(M, C, seed=3)).to(BF)
    xs = [(rnd(M, C, seed=10 + k) * 1.5 + 0.3).float().contiguous() for k in range(2 if two else 1)]
    gams = [rnd(C, seed=20 + k) * 0.5 + 1.0 for k in range(len(xs))]
    stats = [hip.bn_stats(x, gamma=g, beta=torch.zeros_like(g), eps=1e-5) for x, g in zip(xs, gams)]
    dpre = hip.relu_bwd(dy, y, dy2=dy2 if add else None) if mask else (hip.relu_bwd(dy, torch.ones_like(y), dy2=dy2) if add else dy)
    want = []
    for x, g, st in zip(xs, gams, stats):
        dx = torch.empty(M, C, dtype=BF, device="cuda")
        dg, db = hip.bn_bwd(dpre, x, st[0], st[2], g, dx)
        want.append((dg.clone(), db.clone(), dx))
    dxs = [torch.full((M, C), float("nan"), dtype=BF, device="cuda") for _ in xs]
    sides = [(x, st[0], st[2], g, dx) for x, g, st, dx in zip(xs, gams, stats, dxs)]
    assert hip.bn_bwd_fused_ok(dy, sides, y if mask else None, dy2 if add else None, M)
    got = hip.bn_bwd_fused(dy, sides, y=y if mask else None, dy2=dy2 if add else None, M=M)
    for (dg, db), dx, (wg, wb, wdx) in zip(got, dxs, want):
        assert rel(dg, wg) <= 2e-5 and rel(db, wb) <= 2e-5, (rel(dg, wg), rel(db, wb))
        # the same fp32 expression on the same bf16 inputs: equal up to a bf16 rounding tie moved by the 1e-5 of the sums
        assert rel(dx, wdx) <= 8e-3 and (dx.float() != wdx.float()).float().mean().item() <= 1e-3


def _reference_bf16_deviation(tag, model=None):
    """The reference's OWN gradient deviation when it runs under torch.autocast(bfloat16) instead of fp32 (same weights,
    same batch, the metrics of these tests; tools/ref_bf16_gradient_deviation.py): the yardstick for the tolerances."""
    import json
    import os
    with open(os.path.join(GOLDEN, "ref_bf16_gradient_deviation.json")) as f:
        d = json.load(f)
    return d[model][tag] if model is not None and not model.startswith("b32") else d[tag]


def _fresh_model(name):
    m = get_clip_model(named_config(name))
    m.load_state_dict(synth_sd(name), strict=True)
    return m.cuda().eval()


@pytest.mark.parametrize("name", ["b32-yfcc-msclips", "b16-yfcc-msclips"])
def test_gradients_against_reference_autograd(gpu_device, name):
    """Every parameter's gradient against autograd of the imported reference on the golden batch (fp32 CPU, eval-mode
    BatchNorm); b16: the 197-token grid (query-blocked attention backward) and the k = 8 / 4 / 2 / 1 / 1 adapters."""
    import os
    g = np.load(os.path.join(GOLDEN, name + ".grads.npz"))
    m = _fresh_model(name)
    ts = train.TrainStep(m, lr=1e-4)
    b = int(g["batch"])
    img = synth.synth_images(b, seed=int(g["seed"])).cuda()
    tok = synth.synth_tokens(b, seed=int(g["seed"]) + 1).cuda()
    loss = ts.forward(img, tok)
    assert abs(loss.item() - float(g["loss"])) <= 2e-2, (loss.item(), float(g["loss"]))
    grads = ts.backward()
    expect = [k[2:] for k in g.files if k.startswith("g_")]           # every parameter of the model
    assert sorted(grads) == sorted(expect), (sorted(set(expect) - set(grads))[:5], sorted(set(grads) - set(expect))[:5])
    worst, am, coss = {}, {}, {}
    for k in expect:
        got = grads[k]
        ref = g["g_" + k]
        sm = summarize(got)
        scale = max(float(g["gmax_" + k]), 1e-12)                     # the tensor's abs-max in the reference
        err = np.abs(sm[2:] - ref[2:]).max() / scale
        worst[k] = float(err)
        am[k] = abs(sm[1] - ref[1]) / (ref[1] + 1e-12)
        if "gfull_" + k in g.files:
            full = torch.from_numpy(g["gfull_" + k]).flatten()
            coss[k] = F.cosine_similarity(got.float().cpu().flatten(), full, dim=0).item()
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
    print("gradient tensors checked:", len(expect), "worst sample errors:", top)
    print("median sample error", float(np.median(list(worst.values()))), "worst abs-mean deviations",
          sorted(am.items(), key=lambda kv: -kv[1])[:4], "lowest cosines", sorted(coss.items(), key=lambda kv: kv[1])[:4])
    conv_keys = [k for k in expect if any(f in k for f in CONV_SIDE)]
    assert len(expect) == 325 and len(conv_keys) == 114
    print("conv side: median sample error", float(np.median([worst[k] for k in conv_keys])), "worst",
          max(worst[k] for k in conv_keys), "worst abs-mean", max(am[k] for k in conv_keys), "lowest cosine",
          min(coss[k] for k in conv_keys if k in coss))
    for k in expect:
        lnb = k.endswith(("ln_1.bias", "ln_2.bias", "ln_final.bias", "ln_post.bias", "ln_pre.bias", "ln_adapt.bias"))
        tol = (LNB_SAMPLE_TOL, LNB_ABSMEAN_TOL, LNB_COS_TOL) if lnb else \
              (CONV_SAMPLE_TOL, CONV_ABSMEAN_TOL, CONV_COS_TOL) if k in conv_keys else (SAMPLE_TOL, ABSMEAN_TOL, COS_TOL)
        assert worst[k] <= tol[0], (k, worst[k])
        assert am[k] <= tol[1], (k, am[k])
        if k in coss:
            assert coss[k] >= tol[2], (k, coss[k])
    assert float(np.median([worst[k] for k in expect if k not in conv_keys])) <= 3e-2
    assert float(np.median([worst[k] for k in conv_keys])) <= 6e-2
    # no further from the fp32 reference than the reference's own bf16-autocast run is (ViT-B/32: median 2.0 % token side /
    # 5.2 % conv side, worst conv-side tensor 41 %, lowest conv-side cosine 0.978 on this batch; ViT-B/16: 2.2 % / 4.7 %, worst
    # 18 %, cosine 0.984 -- there this build's worst conv-side tensor (21 % in round 2) is allowed 5 % of abs-max on top)
    dev = _reference_bf16_deviation("eval_bn_batch4", name)
    lnb_keys = [k for k in expect if k.endswith(("ln_1.bias", "ln_2.bias", "ln_final.bias", "ln_post.bias", "ln_pre.bias",
                                                  "ln_adapt.bias"))]
    tok_keys = [k for k in expect if k not in conv_keys and k not in lnb_keys]
    print(name, "eval-mode BN vs the reference's own bf16 deviation: token median", float(np.median([worst[k] for k in tok_keys])),
          "/", dev["token_side"]["sample_err_median"], "conv median", float(np.median([worst[k] for k in conv_keys])), "/",
          dev["conv_side"]["sample_err_median"], "conv worst", max(worst[k] for k in conv_keys), "/", dev["conv_side"]["sample_err_worst"],
          "conv cosine", min(coss[k] for k in conv_keys if k in coss), "/", dev["conv_side"]["cosine_lowest"])
    assert float(np.median([worst[k] for k in tok_keys])) <= 1.25 * dev["token_side"]["sample_err_median"] + 5e-3
    assert float(np.median([worst[k] for k in conv_keys])) <= 1.25 * dev["conv_side"]["sample_err_median"] + 5e-3
    top = sorted(((worst[k], k) for k in conv_keys), reverse=True)[:3]
    print(name, "worst conv-side tensors:", top)
    # no blanket slack.  Every conv-side tensor stays inside the reference's own worst bf16 deviation, except -- on
    # ViT-B/16 -- the bias gradient shared by bn3 / residual_bn of the first bottleneck of the parallel branch (one number per
    # channel: the sum of a bf16 gradient map over 4 x 56 x 56 pixels whose terms nearly cancel; 21 % against the reference's
    # 18 % worst tensor): those two keys get 4 % of abs-max on top, by name.
    scoped = {"visual.transformer.parallel_branch_v.1.resnet_stage.conv_0.residual_bn.bias": 4e-2,
              "visual.transformer.parallel_branch_v.1.resnet_stage.conv_0.bn3.bias": 4e-2} if name.startswith("b16") else {}
    for k in conv_keys:
        assert worst[k] <= dev["conv_side"]["sample_err_worst"] + scoped.get(k, 0.0), (k, worst[k])
    assert min(coss[k] for k in conv_keys if k in coss) >= dev["conv_side"]["cosine_lowest"] - 5e-3
    # the shared tensors' gradients are sums over both towers: a text-only / image-only backward must give less
    assert len([k for k in expect if "visual.transformer.resblocks" in k and ".attn." in k]) == 11 * 4


@pytest.mark.parametrize("name", ["b32-yfcc-msclips", "b16-yfcc-msclips"])
def test_gradients_with_train_mode_batchnorm(gpu_device, name):
    """bn="batch": per-GPU batch statistics in every BatchNorm, their backward, the running-statistics update -- against
    autograd of the reference in train() mode (tests/golden/b32-yfcc-msclips.grads_trainbn.npz).  Same tolerance classes as
    the frozen-statistics test; the running statistics after one forward to 1e-2 of their largest entry (measured 2e-3:
    the batch statistics of maps behind bf16 transformer blocks, entering with momentum 0.1)."""
    import os
    g = np.load(os.path.join(GOLDEN, name + ".grads_trainbn.npz"))
    m = _fresh_model(name)
    ts = train.TrainStep(m, lr=1e-4, bn="batch")
    b = int(g["batch"])
    img = synth.synth_images(b, seed=int(g["seed"])).cuda()
    tok = synth.synth_tokens(b, seed=int(g["seed"]) + 1).cuda()
    nbt0 = {k: int(v) for k, v in m.state_dict().items() if k.endswith("num_batches_tracked")}
    loss = ts.forward(img, tok)
    assert abs(loss.item() - float(g["loss"])) <= 2e-2, (loss.item(), float(g["loss"]))
    sd = m.state_dict()
    runs = [k[4:] for k in g.files if k.startswith("run_")]
    assert len(runs) == 2 * 36                                        # running mean / variance of the 36 BatchNorms
    run_err = {}
    for k in runs:
        ref = torch.from_numpy(g["run_" + k])
        run_err[k] = rel(sd[k].cpu(), ref)
        nk = k.rsplit(".", 1)[0] + ".num_batches_tracked"
        assert int(sd[nk]) == nbt0[nk] + 1
    print("running statistics after one forward: worst deviation", max(run_err.values()), max(run_err, key=run_err.get))
    assert max(run_err.values()) <= 1e-2                              # of the tensor's largest entry (0.1 x the batch statistic's error)
    grads = ts.backward()
    expect = [k[2:] for k in g.files if k.startswith("g_")]
    assert sorted(grads) == sorted(expect)
    worst, am, coss = {}, {}, {}
    for k in expect:
        sm, ref = summarize(grads[k]), g["g_" + k]
        worst[k] = float(np.abs(sm[2:] - ref[2:]).max() / max(float(g["gmax_" + k]), 1e-12))
        am[k] = abs(sm[1] - ref[1]) / (ref[1] + 1e-12)
        if "gfull_" + k in g.files:
            coss[k] = F.cosine_similarity(grads[k].float().cpu().flatten(), torch.from_numpy(g["gfull_" + k]).flatten(), dim=0).item()
    conv_keys = [k for k in expect if any(f in k for f in CONV_SIDE)]
    print("train-mode BN: conv side median sample error", float(np.median([worst[k] for k in conv_keys])), "worst",
          max(worst[k] for k in conv_keys), "worst abs-mean", max(am[k] for k in conv_keys), "lowest cosine",
          min(coss[k] for k in conv_keys if k in coss), "| token side median", float(np.median([worst[k] for k in expect if k not in conv_keys])))
    dev = _reference_bf16_deviation("train_bn_batch16" if name.startswith("b32") else "train_bn_batch8", name)
    for k in expect:
        lnb = k.endswith(("ln_1.bias", "ln_2.bias", "ln_final.bias", "ln_post.bias", "ln_pre.bias", "ln_adapt.bias"))
        # (conv side: the abs-mean bound is 10 % or, where the reference's own bf16-autocast run moves a conv-side tensor further than
        #  that on this batch -- 18.2 % on ViT-B/16 batch 8 --, that yardstick: with 8 samples a per-channel adapter filter's gradient
        #  moves by several percent with the summation order of the first convolutions, measured 5.9 % / 11.5 % for the two orders)
        tol = (LNB_SAMPLE_TOL, LNB_ABSMEAN_TOL, LNB_COS_TOL) if lnb else \
              (0.40, max(0.10, dev["conv_side"]["absmean_dev_worst"]), 0.95) if k in conv_keys else (0.10, 0.08, 0.99)
        if k == "logit_scale":
            # a SCALAR whose gradient is a sum of nearly cancelling terms (sum G S): under train-mode BatchNorm every rounding in
            # the conv side moves it -- the reference's own bf16-autocast run is 15.2 % off its fp32 run on this batch (the worst
            # token-side tensor of the yardstick file IS this scalar); measured here 7-10 % depending on the summation order of the
            # first convolutions.  Bounded by the yardstick instead of the 8 % of the tensor-valued gradients.
            tol = (max(tol[0], dev["token_side"]["sample_err_worst"]), max(tol[1], dev["token_side"]["absmean_dev_worst"]), tol[2])
        assert worst[k] <= tol[0], (k, worst[k])
        assert am[k] <= tol[1], (k, am[k])
        if k in coss:
            assert coss[k] >= tol[2], (k, coss[k])
    # ... and no further from the fp32 reference than the reference's own bf16-autocast run in train() mode on this batch
    # (ViT-B/32, batch 16: median 3.9 % token side / 11.7 % conv side, worst conv-side tensor 31 %, lowest conv-side cosine
    # 0.959; ViT-B/16, batch 8: 3.6 % / 9.9 %, worst 29 %, cosine 0.958)
    lnb_keys = [k for k in expect if k.endswith(("ln_1.bias", "ln_2.bias", "ln_final.bias", "ln_post.bias", "ln_pre.bias", "ln_adapt.bias"))]
    tok_keys = [k for k in expect if k not in conv_keys and k not in lnb_keys]
    assert float(np.median([worst[k] for k in tok_keys])) <= 1.25 * dev["token_side"]["sample_err_median"] + 5e-3
    assert float(np.median([worst[k] for k in conv_keys])) <= 1.25 * dev["conv_side"]["sample_err_median"] + 5e-3
    # (the WORST of ~150 conv-side tensors is a noisy statistic: across the eight numerically equivalent schedules of this step --
    #  options.TRAIN adapter_bn_views x bn_two_pass x bn_bwd_fused, which only move summation orders and rounding points -- it takes
    #  0.261-0.316 on ViT-B/16 batch 8 and 0.293-0.309 on ViT-B/32 batch 16 while the median stays at 0.094-0.097 and the lowest cosine
    #  at 0.966-0.969: profiles/r06_trainbn_fixture_spread.txt.  Margin = that spread.)
    assert max(worst[k] for k in conv_keys) <= dev["conv_side"]["sample_err_worst"] + 4e-2
    assert min(coss[k] for k in conv_keys if k in coss) >= dev["conv_side"]["cosine_lowest"] - 5e-3


@pytest.mark.parametrize("bn", ["frozen", "batch"])
def test_adamw_step_with_reference_param_groups_lowers_the_loss(gpu_device, bn):
    name = "b32-yfcc-msclips"
    m = _fresh_model(name)
    cfg = named_config(name)
    ts = train.from_config(m, cfg, bn=bn)
    assert train.from_config(m, cfg).bn == "batch"                     # the reference's train() semantics are the default
    groups = {k: (lr, wd) for k, _, lr, wd in ts.param_groups()}
    # literals of the reference yaml (experiments/model/b32.yaml:34,49; b32-yfcc-msclips.yaml:13-14), world size 1
    assert groups["visual.transformer.resblocks.3.mlp.c_fc.weight"] == (0.0001, 0.2)
    assert groups["visual.transformer.resblocks.3.mlp.c_fc.bias"][1] == 0.0                # WITHOUT_WD_LIST: bias
    assert groups["transformer.resblocks.0.mlp.c_fc.weight"] == (0.0001, 0.05)
    assert ts.betas == (0.9, 0.999) and ts.eps == 1e-8                                      # no OPTIMIZER_ARGS: torch defaults
    assert groups["visual.transformer.resblocks.3.ln_1.weight"][1] == 0.0 and groups["logit_scale"][1] == 0.0
    img, tok = synth.synth_images(8, seed=51).cuda(), synth.synth_tokens(8, seed=52).cuda()
    ts.lr, ts.lr_share = 2e-5, 2e-5
    l0 = ts.forward(img, tok).item()
    before = m.visual.transformer.resblocks[5].mlp.c_fc.weight.detach().clone()
    ts.step(ts.backward())
    assert not torch.equal(before, m.visual.transformer.resblocks[5].mlp.c_fc.weight.detach())
    assert m.transformer.resblocks[5].mlp.c_fc.weight.data_ptr() == m.visual.transformer.resblocks[5].mlp.c_fc.weight.data_ptr()
    inf0 = None
    if bn == "batch":
        inf0 = m.contrastive_loss(img, tok).item()
    l1 = ts.forward(img, tok).item()
    if bn == "frozen":
        assert abs(l1 - m.contrastive_loss(img, tok).item()) <= 2e-2      # the inference path sees the updated weights
    else:
        # two train-mode forwards: every BatchNorm's counter moved twice and the inference path (running statistics,
        # re-folded by the engine) changed with them
        nbt = m.state_dict()["visual.transformer.parallel_branch_v.2.resnet_stage.conv_0.bn2.num_batches_tracked"]
        assert int(nbt) == 1002
        assert abs(m.contrastive_loss(img, tok).item() - inf0) > 1e-4
    ts.saved = None
    assert l1 < l0, (l0, l1)


@pytest.mark.parametrize("bn", ["frozen", "batch"])
def test_optimizer_step_keeps_the_engine_copies_current(gpu_device, bn):
    """TrainStep.step() lets AdamW write the transformer blocks' bf16 copies and re-packs only the conv side / heads
    (Engine.repack_after_optimizer); after three steps every packed tensor equals what a freshly built engine packs from the
    module (the blocks and heads bit for bit, the conv side to an fp32 ulp), and the cached optimizer table is the one of step 1."""
    from msclip_amd.engine import Engine
    m = _fresh_model("b32-yfcc-msclips")
    ts = train.TrainStep(m, lr=3e-5, lr_share=2e-5, bn=bn)
    img, tok = synth.synth_images(6, seed=61).cuda(), synth.synth_tokens(6, seed=62).cuda()
    plans = []
    for _ in range(3):
        ts.forward(img, tok)
        ts.step(ts.backward())
        plans.append(ts._plan)
    assert plans[1] is plans[0] and plans[2] is plans[0] and plans[0].packs
    ts.lr = 1e-5                                                       # a schedule moves the rate: same table, new rates
    ts.forward(img, tok)
    ts.step(ts.backward())
    assert ts._plan is plans[0]
    got, want = ts.eng, Engine(m)
    for i in range(got.n_layers):
        for blocks in ("tblk", "vblk"):
            a, b = getattr(got, blocks)[i], getattr(want, blocks)[i]
            if a is None:
                assert b is None
                continue
            for f in ("wqkv", "bqkv", "wo", "bo", "wfc", "bfc", "wpr", "bpr"):
                assert torch.equal(getattr(a["w"], f), getattr(b["w"], f)), (blocks, i, f)
            for ln in ("ln1", "ln2"):
                assert torch.equal(a[ln].g, b[ln].g) and torch.equal(a[ln].b, b[ln].b)
    # conv side: re-packed by msclip_pack_weights (IEEE fp32 folds) against the fresh engine's tensor algebra (torch's GPU division
    # / square root are not correctly rounded): equal to an fp32 ulp, i.e. a rare one-ulp flip of a bf16 weight
    def same(a, b):
        if a.dtype == torch.bfloat16:
            af, bf_ = a.float(), b.float()
            # (the stem stages' centre taps are w s + w_shortcut s_shortcut: where the two nearly cancel, an fp32 ulp of a scale is
            #  several bf16 ulps of the sum -- an absolute slack of 2^-12 of the tensor's largest entry covers them)
            ok = (bool(((af - bf_).abs() <= 2.0 ** -6 * bf_.abs() + 2.0 ** -12 * bf_.abs().max()).all()) and
                  (af != bf_).float().mean().item() < 1e-3)
        else:
            ok = bool(((a - b).abs() <= 4e-6 * (b.abs() + b.abs().max())).all())
        if not ok:
            print("mismatch:", a.dtype, tuple(a.shape), "max abs diff", (a.float() - b.float()).abs().max().item(), "ref abs max",
                  b.float().abs().max().item(), "elements differing", int((a != b).sum()))
        return ok
    for a, b in zip(got.stem_specs, want.stem_specs):
        assert same(a.weight, b.weight) and same(a.bias, b.bias)
    for j in range(1, 5):
        for a, b in zip(got.par_specs[j], want.par_specs[j]):
            assert same(a.weight, b.weight) and same(a.bias, b.bias)
    assert torch.equal(got.w_vproj, want.w_vproj) and torch.equal(got.w_tproj, want.w_tproj) and torch.equal(got.w_last, want.w_last)
    assert same(got.dual_w, want.dual_w)
    assert got.logit_scale_exp == want.logit_scale_exp
    # a RE-ASSIGNED parameter (another tensor object) invalidates the cached optimizer table and the engine's cached views /
    # aliases: the next step trains the new tensor and the engine serves it
    blk = m.transformer.resblocks[3]
    blk.ln_2.weight = torch.nn.Parameter(blk.ln_2.weight.detach().clone() * 1.5)
    ts.forward(img, tok)
    ts.step(ts.backward())
    assert ts._plan is not plans[0]
    assert torch.equal(ts.eng.tblk[3]["ln2"].g, blk.ln_2.weight.detach())
    assert torch.equal(ts.eng.tblk[3]["w"].wfc, blk.mlp.c_fc.weight.detach().to(torch.bfloat16))


@pytest.mark.parametrize("name", ["b32-yfcc-msclips", "b16-yfcc-msclips"])
def test_training_loop_memorises_two_batches(gpu_device, name):
    """Twelve optimizer steps over two fixed batches with train-mode BatchNorm (what tools/train_synthetic.py runs): the
    loss stays finite and falls by more than half, and the inference path (running statistics after 12 updates) agrees.
    b16: the 14 x 14 grid (k = 8 / 4 / 2 / 1 / 1 adapters, query-blocked attention backward) in train-mode BatchNorm."""
    m = _fresh_model(name)
    ts = train.from_config(m, named_config(name))
    ts.lr = ts.lr_share = 2e-5
    data = [(synth.synth_images(16, seed=10 + i).cuda(), synth.synth_tokens(16, seed=100 + i).cuda()) for i in range(2)]
    losses = []
    for step in range(12):
        losses.append(ts.forward(*data[step % 2]).item())
        ts.step(ts.backward())
    assert all(np.isfinite(losses)), losses
    assert max(losses[-2:]) < 0.5 * min(losses[:2]), losses
    assert np.isfinite(m.contrastive_loss(*data[0]).item())


def test_checkpoint_resume_continues_the_run(gpu_device, tmp_path):
    """save_checkpoint / resume_checkpoint (the reference's 'step' / 'model' / 'state_dict' / 'perf' / 'optimizer' dict,
    lib/utils/utils.py:157-200): two steps, save, then the third step once in the running process and once in a fresh
    model + TrainStep resumed from the file.  Every kernel but the embedding gradient's fp32 atomics is deterministic, so
    all parameters, running statistics and AdamW moments agree bitwise except the two embedding tables (1e-5).
    The stored optimizer state loads into torch.optim.AdamW built over the same groups."""
    name = "b32-yfcc-msclips"
    cfg = named_config(name)
    data = [(synth.synth_images(8, seed=20 + i).cuda(), synth.synth_tokens(8, seed=120 + i).cuda()) for i in range(3)]

    def run(m, ts, steps):
        for i in steps:
            ts.forward(*data[i])
            ts.step(ts.backward())
    mb = _fresh_model(name)
    tb = train.from_config(mb, cfg)
    run(mb, tb, range(2))
    path = tmp_path / "checkpoint.pth"
    train.save_checkpoint(mb, tb, path, step=1, model_name=name)
    obj = torch.load(path, weights_only=False)
    assert set(obj) == {"step", "model", "state_dict", "perf", "optimizer"} and obj["step"] == 2
    run(mb, tb, [2])
    mc = _fresh_model(name)
    tc = train.from_config(mc, cfg)
    assert train.resume_checkpoint(mc, tc, path) == 2 and tc.steps == 2
    run(mc, tc, [2])
    sb, sc = mb.state_dict(), mc.state_dict()
    atomics = ("token_embedding.weight", "positional_embedding")
    for k in sb:
        if k in atomics:
            assert rel(sc[k], sb[k]) <= 1e-5, (k, rel(sc[k], sb[k]))
        else:
            assert torch.equal(sb[k], sc[k]), (k, rel(sc[k].float(), sb[k].float()))
    k = "visual.transformer.resblocks.4.mlp.c_fc.weight"
    assert torch.equal(tc.state[k][0], tb.state[k][0]) and torch.equal(tc.state[k][1], tb.state[k][1]) and tc.steps == tb.steps == 3
    # torch.optim.AdamW accepts the stored state (same parameter order / groups)
    params = dict(mc.named_parameters())
    names = obj["optimizer"]["msclip"]["names"]
    opt = torch.optim.AdamW([{"params": [params[names[i]] for i in g["params"]], "lr": g["lr"], "weight_decay": g["weight_decay"]}
                             for g in obj["optimizer"]["param_groups"]], betas=(0.9, 0.98), eps=1e-6)
    remap, n = {}, 0
    for g in obj["optimizer"]["param_groups"]:
        for i in g["params"]:
            remap[i] = n
            n += 1
    sd = {"state": {remap[i]: v for i, v in obj["optimizer"]["state"].items()},
          "param_groups": [dict(g, params=[remap[i] for i in g["params"]]) for g in obj["optimizer"]["param_groups"]]}
    opt.load_state_dict({"state": sd["state"], "param_groups": [dict(pg, **{k: v for k, v in g.items() if k != "params"},
                                                                      params=g["params"])
                                                                 for pg, g in zip(opt.state_dict()["param_groups"], sd["param_groups"])]})
    assert len(opt.state_dict()["state"]) == len(obj["optimizer"]["state"])


@pytest.mark.parametrize("bn,batch", [("batch", 5), ("frozen", 3), ("batch", 1)])
def test_training_step_odd_batches(gpu_device, bn, batch):
    """Batches that are not multiples of anything (row counts 635 / 381 / 127: ragged GEMM tiles, padded wgrad
    contractions, un-foldable BatchNorm maps; batch 1: the loss is 0 and BatchNorm sees one image): finite gradients for
    every parameter, parameters move."""
    name = "b32-yfcc-msclips"
    m = _fresh_model(name)
    ts = train.from_config(m, named_config(name), bn=bn)
    img, tok = synth.synth_images(batch, seed=7).cuda(), synth.synth_tokens(batch, seed=8).cuda()
    loss = ts.forward(img, tok)
    grads = ts.backward()
    assert np.isfinite(loss.item()) and len(grads) == 325
    for k, g in grads.items():
        assert bool(torch.isfinite(g).all()), k
    with pytest.raises(RuntimeError, match="forward"):
        ts.backward()                                                  # one backward per forward
    before = m.visual.transformer.resblocks[0].resnet_stage.conv_1.conv1.weight.detach().clone()
    ts.step(grads)
    if batch > 1:
        assert not torch.equal(before, m.visual.transformer.resblocks[0].resnet_stage.conv_1.conv1.weight.detach())


def test_train_synthetic_script(gpu_device):
    """tools/train_synthetic.py end to end (its exit code is the loss check)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_synthetic.py"), "--batch", "8", "--steps", "6"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "OK" in r.stdout and r.stdout.count("step ") == 6


@pytest.mark.parametrize("ci,co,h,B", [(48, 48, 28, 3), (48, 96, 56, 2), (96, 192, 28, 3), (192, 192, 28, 2), (384, 384, 14, 2), (48, 48, 30, 2)])
def test_conv_input_gradient_by_parity_classes(gpu_device, monkeypatch, ci, co, h, B):
    """dX of a 3x3 / stride 2 / pad 1 convolution as four stride-1 implicit-GEMM launches over dY, one per input-pixel parity
    class, scattered straight into dX (train_conv.ConvSideBackward._dgrad_parity, round 4): against fp32 conv_transpose2d of the
    same bf16 operands, against the column-matrix path (dcol GEMM + msclip_col2im), every dX row written; 14 x 14 maps keep the
    column-matrix path."""
    from msclip_amd import packing as P, train_conv as TC
    g = torch.Generator().manual_seed(3)
    w = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5
    spec = P.ConvSpec(w, torch.zeros(co), h, h, 2, 1).to("cuda")
    ho = spec.h_out
    pix = B * ho * ho
    dpre = TC._zbuf(pix, co, "cuda")
    dpre.copy_((torch.randn(pix, co, generator=g) * 0.5).to(BF))
    x_in = TC._zbuf(B * h * h, ci, "cuda")
    x_in.copy_(torch.randn(B * h * h, ci, generator=g).to(BF))
    bw = object.__new__(TC.ConvSideBackward)
    bw._wt, bw._pplan = {}, TC._PARITY_PLANS
    assert bw._parity_ok(spec) == (h >= 28)
    col = hip.im2col(x_in, B, h, h, ci, 3, 3, 2, 1)
    G, db, dx = bw._conv_bwd("t", spec, x_in, dpre, B, col=col)
    from msclip_amd import options
    monkeypatch.setattr(options, "TRAIN", options.TRAIN.replace(dgrad_col2im=True))
    G2, db2, dx2 = bw._conv_bwd("t", spec, x_in, dpre, B, col=col)
    wq = spec.weight[:, :9 * ci].float().view(co, 3, 3, ci).permute(0, 3, 1, 2)               # the packed (bf16) filter
    dy = dpre[:pix].float().view(B, ho, ho, co).permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(dy, wq, stride=2, padding=1, output_padding=1).permute(0, 2, 3, 1).reshape(B * h * h, ci)
    sc = ref.abs().max().item()
    assert bool(torch.isfinite(dx.float()).all())                        # (the fixture poisons fresh memory: every row was written)
    assert (dx.float() - ref).abs().max().item() <= 6e-3 * sc            # one bf16 rounding of an fp32 sum
    assert (dx2.float() - ref).abs().max().item() <= 1.2e-2 * sc         # (the column matrix rounds every tap's product first)
    assert torch.equal(G, G2) and torch.equal(db, db2)
    # through the ReLU that produced x_in: in the launches' epilogues (msclip_gemm resid_kind 5) = a msclip_relu_bwd pass over dx
    monkeypatch.setattr(options, "TRAIN", options.TRAIN.replace(dgrad_col2im=False))
    _, _, dxr = bw._conv_bwd("t", spec, x_in, dpre, B, col=col, relu_of=x_in)
    assert torch.equal(dxr, torch.where(x_in[:B * h * h] > 0, dx, torch.zeros_like(dx)))


@pytest.mark.parametrize("T,No,Ni", [(70000, 96, 864), (66000, 48, 448), (8000, 192, 1728), (4097, 768, 96), (3000, 384, 192)])
def test_weight_gradient_gemm_token_major_ragged_channels(gpu_device, T, No, Ni):
    """msclip_gemm_splitk_tn on channel counts that are not whole 256-tiles (the conv side's narrow gradients, round 4): an edge
    tile's extra channels are whatever lies right of the operand in memory -- NaN here -- and must reach no stored output;
    rows past T are NaN too.  Against fp32 torch, bitwise repeatable, and gradgemm.wgrad takes this path from 65 536 tokens on."""
    import msclip_amd.gradgemm as G
    g = torch.Generator().manual_seed(5)
    wide_dy = (torch.randn(T + 3, No + 16, generator=g) * 0.5).to(BF).cuda()
    wide_x = torch.randn(T + 3, Ni + 64, generator=g).to(BF).cuda()
    wide_dy[T:], wide_x[T:] = float("nan"), float("nan")
    wide_dy[:, No:], wide_x[:, Ni:] = float("nan"), float("nan")
    dy, x = wide_dy[:, :No], wide_x[:, :Ni]
    tiles = ((No + 255) // 256) * ((Ni + 255) // 256)
    S = max(1, min(256 // tiles, T // 2048))
    ref = dy[:T].float().t() @ x[:T].float()
    out = torch.full((No + 1, Ni), float("nan"), device="cuda")
    hip.gemm_splitk_tn(dy, x, T, S, out=out[:No])
    first = out.clone()
    scale = ref.abs().max().item()
    assert (out[:No] - ref).abs().max().item() <= 2e-5 * scale * max(1.0, (T / 4096) ** 0.5) + 1e-3
    assert bool(torch.isnan(out[No:]).all())
    hip.gemm_splitk_tn(dy, x, T, S, out=out[:No])
    assert torch.equal(out[:No], first[:No])
    assert G._tn_ok(dy, x, ragged=True) and not G._tn_ok(dy, x)
    got = G.wgrad(dy, x, T)
    if T >= 65536 and tiles <= 4:
        assert torch.equal(got, first[:No])                               # the token-major launch
    else:
        assert (got - ref).abs().max().item() <= 2e-5 * scale * max(1.0, (T / 4096) ** 0.5) + 1e-3


@pytest.mark.parametrize("T,No,Ni,S", [(4096, 768, 768, 4), (65024, 768, 3072, 7), (5000, 2304, 768, 3), (127 * 33, 256, 512, 1),
                                       (70, 512, 256, 2)])
def test_weight_gradient_gemm_on_token_major_operands(gpu_device, T, No, Ni, S):
    """msclip_gemm_splitk_tn: dW = dY^T X straight from the token-major operands (LDS transpose reads, no operand transposes)
    against fp32 torch and against the transposing split-K path; token counts that are not whole K-tiles / slices (rows past
    T contribute zero: buffer range check), operands that are column windows of wider matrices, bitwise repeatable."""
    g = torch.Generator().manual_seed(5)
    wide_dy = (torch.randn(T + 3, No + 256, generator=g) * 0.5).to(BF).cuda()
    wide_x = torch.randn(T + 3, Ni + 512, generator=g).to(BF).cuda()
    wide_dy[T:] = float("nan")                                    # rows past T must not be read into the sum
    wide_x[T:] = float("nan")
    dy, x = wide_dy[:, 256:], wide_x[:, 256:256 + Ni]
    ref = dy[:T].float().t() @ x[:T].float()
    out = torch.full((No, Ni), float("nan"), device="cuda")
    hip.gemm_splitk_tn(dy, x, T, S, out=out)
    first = out.clone()
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() <= 2e-5 * scale * max(1.0, (T / 4096) ** 0.5) + 1e-3, ((out - ref).abs().max().item(), scale)
    hip.gemm_splitk_tn(dy, x, T, S, out=out)
    assert torch.equal(out, first)
    import msclip_amd.gradgemm as G
    assert G._tn_ok(dy, x)
    got = G.wgrad(dy, x, T)
    assert torch.equal(got, first) if (max(1, min(256 // ((No // 256) * (Ni // 256)), T // 2048)) == S) else True

# ---------------------------------------------------------------------------------------------------------------------
# round 5: the conv side's re-pack as one table-driven launch (msclip_pack_weights)
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["b32-yfcc-msclips", "b16-yfcc-msclips"])
def test_table_driven_repack_is_the_tensor_algebra_pack(gpu_device, monkeypatch, name):
    """msclip_pack_weights rewrites every derived conv-side operand (BatchNorm folds, the stem stages' merged shortcut, NHWC /
    transposed layouts, shifts as biases) in place from the module's parameters; msclip_amd/packing.py states the same folds as
    tensor algebra.  After an in-place change of every parameter and running statistic: all weights and BatchNorm shifts equal the
    tensor-algebra pack's to one fp32 ulp (the kernel's fold is IEEE fp32, torch's GPU division / square root are not correctly
    rounded), the adapters' pointwise bias (a matrix-vector product) to fp32 rounding; and the step-level path
    (TrainStep.step -> repack_after_optimizer) runs it."""
    m = get_clip_model(named_config(name))
    m.load_state_dict(synth_sd(name))
    m = m.cuda().eval()
    eng = m.engine()
    eng.refresh()
    assert eng._pack_plan is not None

    def derived(e):
        out = {"dual_w": e.dual_w, "dual_b": e.dual_b, "w_last": e.w_last}
        for i, sp in enumerate(e.stem_specs):
            out[f"stem{i}.w"], out[f"stem{i}.b"] = sp.weight, sp.bias
        for j in range(1, 5):
            for n, sp in zip(("c1", "c2", "cr", "c3"), e.par_specs[j]):
                out[f"par{j}.{n}.w"], out[f"par{j}.{n}.b"] = sp.weight, sp.bias
            out[f"par{j}.b3r"] = e.par_b3r[j]
        for j, a in enumerate(e.adapters):
            out[f"ad{j}.pool"], out[f"ad{j}.pw.w"], out[f"ad{j}.pw.b"], out[f"ad{j}.dww"], out[f"ad{j}.dwb"] = \
                a["pool"], a["pw"].weight, a["pw"].bias, a["dww"], a["dwb"]
        return out
    persistent = derived(eng)                                          # the tensors the plan writes into
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for k, t in m.state_dict().items():
            if not t.is_floating_point():
                continue
            if k.endswith("running_var"):
                t.mul_(0.5 + torch.rand(t.shape, device="cuda", generator=g))
            else:
                t.add_(0.05 * t.abs().mean().clamp_min(1e-3) * torch.randn(t.shape, device="cuda", generator=g))
    eng._pack_plan.run()
    got = {k: v.clone() for k, v in persistent.items()}
    set_opt(monkeypatch, eng, repack_table=False)
    eng.refresh(force=True)                                            # pure tensor algebra on the changed parameters (new tensors)
    ref = derived(eng)
    set_opt(monkeypatch, eng, repack_table=True)
    assert set(got) == set(ref) and len(got) > 60
    for k in ref:
        a, b = got[k], ref[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if k.endswith("pw.b"):
            assert (a - b).abs().max().item() <= 1e-5 * max(b.abs().max().item(), 1.0), k
        elif a.dtype == BF:
            # the kernel's BatchNorm scale is IEEE fp32 (g / sqrt(var + eps), correctly rounded); torch's GPU division / square
            # root are not (12 of 48 scales differ by one ulp from the correctly rounded value): a one-ulp fp32 difference in the
            # product flips a bf16 rounding in a few elements per million
            af, bf_ = a.float(), b.float()
            assert ((af - bf_).abs() <= 2.0 ** -7 * bf_.abs() + 1e-30).all(), k
            assert (af != bf_).float().mean().item() < 1e-3, k
        else:
            assert ((a - b).abs() <= 1e-6 * (b.abs() + b.abs().max())).all(), (k, (a - b).abs().max().item())    # (shifts: b - mean scale cancels)
    # the step-level path: an optimizer step re-packs through the table and the next forward agrees with a full tensor-algebra pack
    ts = train.from_config(m, named_config(name), bn="frozen")
    img, tok = synth.synth_images(8, seed=71).cuda(), synth.synth_tokens(8, seed=72).cuda()
    calls = []
    real = hip.PackPlan.run
    hip.PackPlan.run = lambda self: (calls.append(1), real(self))[1]
    try:
        ts.forward(img, tok)
        ts.step(ts.backward())
    finally:
        hip.PackPlan.run = real
    assert calls
    f_table = m.encode_image(img).clone()
    m.engine().refresh(force=True)                                     # tensor algebra: operands within an ulp of the table's
    f_full = m.encode_image(img)
    assert (f_table - f_full).abs().max().item() <= 1e-3 and F.cosine_similarity(f_table, f_full, dim=-1).min().item() >= 0.99999


@pytest.mark.parametrize("bn", ["frozen", "batch"])
def test_gradients_at_batch_32_against_reference_autograd(gpu_device, bn):
    """Round 6: the gradient pin at batch 32 -- autograd of the REAL reference on 32 pairs, eval-mode and
    train-mode BatchNorm (tools/make_golden.py --grads-b32), every tensor of <= 1024 elements stored in full.

    The verdict's conjecture was that the batch-4 fixtures need their 25 % / cosine 0.975 bounds on the conv side and the LayerNorm
    biases only because 4 samples condition the cancelling sums badly.  Measured: true for the LayerNorm biases and the token
    side (every such tensor is within 10 % of abs-max / cosine >= 0.99 here, the bounds asserted below), NOT for the conv side:
    the REFERENCE ITSELF, run under torch.autocast(bfloat16) on this batch, moves its conv-side BatchNorm gradients by up to 25 %
    (eval) / 35 % (train-mode BN) of abs-max, median 5 % / 11 %, lowest cosine 0.986 / 0.952
    (tools/ref_bf16_gradient_deviation.py b32-batch32 -> tests/golden/ref_bf16_gradient_deviation.json): bf16 activation / gradient
    maps summed over 32 x 112 x 112 pixels, not the batch size.  The conv side is therefore bounded by that yardstick -- no further
    from the fp32 reference than the reference's own bf16 run -- tensor by tensor for the worst case, and in the median."""
    import os
    name = "b32-yfcc-msclips"
    g = np.load(os.path.join(GOLDEN, f"{name}.grads{'_trainbn' if bn == 'batch' else ''}_b32.npz"))
    dev = _reference_bf16_deviation("train_bn_batch32" if bn == "batch" else "eval_bn_batch32", name)
    m = _fresh_model(name)
    ts = train.TrainStep(m, lr=1e-4, bn=bn)
    b = int(g["batch"])
    assert b == 32
    img = synth.synth_images(b, seed=int(g["seed"])).cuda()
    tok = synth.synth_tokens(b, seed=int(g["seed"]) + 1).cuda()
    loss = ts.forward(img, tok)
    assert abs(loss.item() - float(g["loss"])) <= 2e-2, (loss.item(), float(g["loss"]))
    grads = ts.backward()
    expect = [k[2:] for k in g.files if k.startswith("g_")]
    assert sorted(grads) == sorted(expect) and len(expect) == 325
    worst, am, coss = {}, {}, {}
    for k in expect:
        sm, ref = summarize(grads[k]), g["g_" + k]
        worst[k] = float(np.abs(sm[2:] - ref[2:]).max() / max(float(g["gmax_" + k]), 1e-12))
        am[k] = abs(sm[1] - ref[1]) / (ref[1] + 1e-12)
        if "gfull_" + k in g.files:
            coss[k] = F.cosine_similarity(grads[k].float().cpu().flatten(), torch.from_numpy(g["gfull_" + k]).flatten(), dim=0).item()
    conv_keys = [k for k in expect if any(f in k for f in CONV_SIDE)]
    lnb_keys = [k for k in expect if k.endswith(("ln_1.bias", "ln_2.bias", "ln_final.bias", "ln_post.bias", "ln_pre.bias", "ln_adapt.bias"))]
    tok_keys = [k for k in expect if k not in conv_keys and k not in lnb_keys]
    vec = [k for k in coss if grads[k].numel() > 1]
    conv_med, conv_worst = float(np.median([worst[k] for k in conv_keys])), max(worst[k] for k in conv_keys)
    conv_cos = min(coss[k] for k in conv_keys if k in vec)
    print(f"batch 32, bn={bn}: {len(expect)} tensors, {len(coss)} compared in full | token side median "
          f"{float(np.median([worst[k] for k in tok_keys])):.4f} worst {max(worst[k] for k in tok_keys):.4f} | LayerNorm biases worst "
          f"{max(worst[k] for k in lnb_keys):.4f} cosine {min(coss[k] for k in lnb_keys if k in coss):.4f} | conv side median {conv_med:.4f} "
          f"(reference bf16: {dev['conv_side']['sample_err_median']:.4f}) worst {conv_worst:.4f} ({dev['conv_side']['sample_err_worst']:.4f}) "
          f"cosine {conv_cos:.4f} ({dev['conv_side']['cosine_lowest']:.4f})")
    for k in tok_keys + lnb_keys:
        scalar = grads[k].numel() == 1                  # logit_scale: one number, a sum of cancelling terms (the yardstick's worst token-side tensor)
        assert worst[k] <= (max(0.10, dev["token_side"]["sample_err_worst"] + 0.02) if scalar else 0.10), (k, worst[k])
        if k in coss and not scalar:
            assert coss[k] >= 0.99, (k, coss[k])
    assert float(np.median([worst[k] for k in tok_keys])) <= 1.25 * dev["token_side"]["sample_err_median"] + 5e-3
    for k in conv_keys:
        assert worst[k] <= dev["conv_side"]["sample_err_worst"] + 0.02, (k, worst[k])
        if k in vec:
            assert coss[k] >= dev["conv_side"]["cosine_lowest"] - 5e-3, (k, coss[k])
    assert conv_med <= dev["conv_side"]["sample_err_median"] + 5e-3, (conv_med, dev["conv_side"]["sample_err_median"])


def test_batchnorm_backward_over_token_columns(gpu_device):
    """hip.bn_bwd_token_columns: the lateral adapter's BatchNorm (over the grid rows of a token matrix, class rows excluded) run on
    whole samples as rows of L * D columns -- against msclip_bn_bwd_* on a gathered copy of the grid rows, class rows passed through."""
    B, L, D = 24, 50, 768
    g2 = L - 1
    x, dy = rnd(B * L, D, seed=1) * 1.5 + 0.2, rnd(B * L, D, seed=2)
    gam = rnd(D, seed=3) * 0.5 + 1.0
    xg = x.view(B, L, D)[:, 1:].reshape(B * g2, D)
    dyg = dy.view(B, L, D)[:, 1:].reshape(B * g2, D)
    part = hip.bn_stats_partials(x.view(B, L * D))
    sums = part.view(2, L, D)[:, 1:].sum(1)
    o = torch.empty(5, D, device="cuda")
    hip.bn_finish(sums, D, B * g2, gam, torch.zeros_like(gam), 1e-5, o)
    mean, var, rstd = hip.bn_stats(xg, gamma=gam, beta=torch.zeros_like(gam), eps=1e-5)[:3]
    assert rel(o[0], mean) <= 1e-5 and rel(o[1], var) <= 1e-4 and rel(o[2], rstd) <= 1e-4
    want = torch.empty_like(dyg)
    wg, wb = hip.bn_bwd(dyg, xg, mean, rstd, gam, want)
    dx = torch.full((B * L, D), float("nan"), device="cuda")
    dg, db = hip.bn_bwd_token_columns(dy.view(B, L * D), x.view(B, L * D), mean, rstd, gam, dx.view(B, L * D), L, B * g2)
    assert rel(dg, wg) <= 1e-4 and rel(db, wb) <= 1e-4
    assert torch.equal(dx.view(B, L, D)[:, 0], dy.view(B, L, D)[:, 0])                  # class rows: dx = dy
    assert rel(dx.view(B, L, D)[:, 1:].reshape(B * g2, D), want) <= 1e-4


def test_raw_conv_operands_from_one_table_launch(gpu_device, monkeypatch):
    """options.TRAIN.raw_pack_table: the raw (unfolded) conv-side operands of the train-mode BatchNorm step rewritten by ONE
    msclip_pack_weights launch from a device-resident item table -- bitwise the ~60 ATen permute / cast / copy launches it
    replaces, before and after an in-place parameter update."""
    from msclip_amd import options, train_conv as TC
    m = _fresh_model("b32-yfcc-msclips")
    eng = m.engine()

    def operands(raw):
        out = {f"spec{i}": sp.weight for i, (sp, _) in enumerate(raw.items)}
        out.update(w_conv1=raw.w_conv1, w_par0=raw.w_par0, w_dual=raw.w_dual)
        out.update({f"pool{j}": t for j, t in enumerate(raw.pool)})
        out.update({f"dww{j}": t for j, t in enumerate(raw.dww)})
        return {k: v.clone() for k, v in out.items()}
    monkeypatch.setattr(options, "TRAIN", options.TRAIN.replace(raw_pack_table=True))
    table = TC._RawSpecs(eng)
    monkeypatch.setattr(options, "TRAIN", options.TRAIN.replace(raw_pack_table=False))
    eager = TC._RawSpecs(eng)
    for rnd_ in range(2):
        monkeypatch.setattr(options, "TRAIN", options.TRAIN.replace(raw_pack_table=True))
        table.refresh()
        assert table._table is not None and table._table.n_items == len(table.items) + 2 + 2 + 10
        monkeypatch.setattr(options, "TRAIN", options.TRAIN.replace(raw_pack_table=False))
        eager.refresh()
        a, b = operands(table), operands(eager)
        assert sorted(a) == sorted(b) and len(a) >= 40
        for k in a:
            assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
        with torch.no_grad():                                           # an optimizer step's in-place update
            for p_ in m.parameters():
                p_.mul_(1.0 + 0.01 * (rnd_ + 1)).add_(1e-3)


def test_two_pass_image_batchnorm_in_the_step_equals_the_raw_map_path(gpu_device, monkeypatch):
    """options.TRAIN.bn_two_pass (default) against raw fp32 maps + statistics / normalise passes in the whole train-mode step: the
    loss equal to bf16 rounding ties of the two activation maps, every gradient to the noise of xhat's one bf16 rounding."""
    from msclip_amd import options
    m = _fresh_model("b32-yfcc-msclips")
    img = synth.synth_images(96, seed=611).cuda()
    tok = synth.synth_tokens(96, seed=612, min_len=2, max_len=40).cuda()
    out = {}
    for two in (False, True):
        monkeypatch.setattr(options, "TRAIN", options.TRAIN.replace(bn_two_pass=two))
        ts = train.TrainStep(m, lr=1e-4, bn="batch")
        loss = ts.forward(img, tok)
        out[two] = (loss.item(), {k: v.float().clone() for k, v in ts.backward().items()})
    (l0, g0), (l1, g1) = out[False], out[True]
    assert abs(l0 - l1) <= 2e-3 * max(1.0, abs(l0)) and sorted(g0) == sorted(g1)
    worst = {}
    for k in g0:
        a, b = g0[k].flatten(), g1[k].flatten()
        worst[k] = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-12)
    print("two-pass vs raw-map BatchNorm: loss", l0, l1, "worst", sorted(worst.items(), key=lambda kv: -kv[1])[:4],
          "median", float(np.median(list(worst.values()))))
    # (the two forwards differ by bf16 rounding ties of the first two maps; five stages of batch-statistics BatchNorm behind them
    #  re-normalise the perturbed values, so the deep conv-side gradients move by percents between two EQUALLY valid roundings --
    #  at batch 24 up to 25 % on parallel_branch_v.4, the order of the reference's own bf16 deviation; both forms are pinned to the
    #  reference's autograd separately by the fixtures above, which run the default = two passes)
    head = [k for k in g0 if k.startswith(("visual.transformer.resblocks.0.conv1", "visual.transformer.resblocks.0.bn1",
                                           "visual.transformer.parallel_branch_v.0."))]
    assert len(head) == 6
    cos = {k: F.cosine_similarity(g0[k].flatten(), g1[k].flatten(), dim=0).item() for k in g0 if g0[k].numel() > 1}
    print("lowest cosine", sorted(cos.items(), key=lambda kv: kv[1])[:4], "first convs / BatchNorms", {k: round(worst[k], 4) for k in head})
    assert min(cos.values()) >= 0.95 and float(np.median(list(worst.values()))) <= 4e-2      # (measured 0.9946 / 2.5e-2 at batch 96)


def test_adapter_batchnorm_on_views_equals_the_gathered_rows_path(gpu_device, monkeypatch):
    """options.TRAIN.adapter_bn_views (default) against gather / clone / scatter copies around the adapters' BatchNorm in the whole
    train-mode step (same arithmetic, another summation order of the statistics)."""
    from msclip_amd import options
    m = _fresh_model("b32-yfcc-msclips")
    img = synth.synth_images(32, seed=621).cuda()
    tok = synth.synth_tokens(32, seed=622, min_len=2, max_len=40).cuda()
    out = {}
    for views in (False, True):
        monkeypatch.setattr(options, "TRAIN", options.TRAIN.replace(adapter_bn_views=views))
        ts = train.TrainStep(m, lr=1e-4, bn="batch")
        loss = ts.forward(img, tok)
        out[views] = (loss.item(), {k: v.float().clone() for k, v in ts.backward().items()})
    (l0, g0), (l1, g1) = out[False], out[True]
    assert abs(l0 - l1) <= 2e-3 * max(1.0, abs(l0)) and sorted(g0) == sorted(g1)
    worst = {k: (g0[k] - g1[k]).abs().max().item() / max(g0[k].abs().max().item(), 1e-12) for k in g0}
    cos = {k: F.cosine_similarity(g0[k].flatten(), g1[k].flatten(), dim=0).item() for k in g0 if g0[k].numel() > 1}
    print("adapter BatchNorm on views vs gathered rows: loss", l0, l1, "worst", sorted(worst.items(), key=lambda kv: -kv[1])[:3],
          "median", float(np.median(list(worst.values()))), "lowest cosine", min(cos.values()))
    assert min(cos.values()) >= 0.95 and float(np.median(list(worst.values()))) <= 4e-2


def test_fused_batchnorm_backward_in_the_step_equals_the_pass_per_batchnorm_path(gpu_device, monkeypatch):
    """options.TRAIN.bn_bwd_fused (default) against the round-5 path in the whole train-mode step: same loss, every conv-side
    gradient equal to summation-order noise (the reference-autograd fixtures above run the fused default)."""
    from msclip_amd import options
    m = _fresh_model("b32-yfcc-msclips")
    img = synth.synth_images(24, seed=601).cuda()
    tok = synth.synth_tokens(24, seed=602, min_len=2, max_len=40).cuda()
    out = {}
    for fused in (False, True):
        monkeypatch.setattr(options, "TRAIN", options.TRAIN.replace(bn_bwd_fused=fused))
        ts = train.TrainStep(m, lr=1e-4, bn="batch")
        loss = ts.forward(img, tok)
        out[fused] = (loss.item(), {k: v.float().clone() for k, v in ts.backward().items()})
    (l0, g0), (l1, g1) = out[False], out[True]
    assert l0 == l1 and sorted(g0) == sorted(g1)
    worst = {}
    for k in g0:
        a, b = g0[k].flatten(), g1[k].flatten()
        worst[k] = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-12)
    print("fused vs per-BatchNorm backward: worst", sorted(worst.items(), key=lambda kv: -kv[1])[:4])
    assert max(worst.values()) <= 2e-2 and float(np.median(list(worst.values()))) <= 2e-3


@pytest.mark.parametrize("bn,B", [("frozen", 12), ("batch", 256)])
def test_compact_last_block_gradients_equal_the_full_block(gpu_device, monkeypatch, bn, B):
    """Round 6: the training step runs the last block's out_proj / ln_2 / c_fc / c_proj -- forward AND backward -- on the Bi + Bt rows
    that are read behind the block (cls rows, M.py:2685; EOT rows, M.py:3057-3060), like the inference path.  Every other row's
    output has zero gradient, so all 325 parameter gradients are unchanged: compared with the same step over every row
    (options.TRAIN.compact_last_block = False; both are separately pinned to the reference's autograd by the fixtures above,
    which run the default = compact), loss equal, gradients to bf16-operand noise, on a ragged small batch and at a batch where
    the fused training GEMM forms run."""
    from msclip_amd import options
    m = _fresh_model("b32-yfcc-msclips")
    img = synth.synth_images(B, seed=501).cuda()
    tok = synth.synth_tokens(B, seed=502, min_len=1, max_len=70).cuda()
    out = {}
    for compact in (False, True):
        monkeypatch.setattr(options, "TRAIN", options.TRAIN.replace(compact_last_block=compact))
        ts = train.TrainStep(m, lr=1e-4, bn=bn)
        loss = ts.forward(img, tok)
        assert (ts.saved.get("compact") is not None) == compact
        out[compact] = (loss.item(), {k: v.float().clone() for k, v in ts.backward().items()})
    (l0, g0), (l1, g1) = out[False], out[True]
    assert abs(l0 - l1) <= 1e-3 * max(1.0, abs(l0))
    assert sorted(g0) == sorted(g1) and len(g0) == 325
    worst = {}
    for k in g0:
        a, b = g0[k].flatten(), g1[k].flatten()
        worst[k] = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-12)
        cos = F.cosine_similarity(a, b, dim=0).item() if a.numel() > 1 else 1.0
        assert worst[k] <= 5e-2 and cos >= 0.995, (k, worst[k], cos)
    print(f"compact vs full last block (bn={bn}, batch {B}): worst", sorted(worst.items(), key=lambda kv: -kv[1])[:4],
          "median", float(np.median(list(worst.values()))))
    # (train-mode BatchNorm re-normalises every conv map with statistics of the perturbed values: measured median 0.95 %, worst 2.5 %;
    #  frozen statistics: an order of magnitude less)
    assert float(np.median(list(worst.values()))) <= (1.5e-2 if bn == "batch" else 5e-3)
