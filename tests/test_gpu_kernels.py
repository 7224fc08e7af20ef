"""Per-kernel parity on a real MI355X: each C-ABI entry point against a plain fp32 PyTorch statement of the same op,
on seeded random (asymmetric, transpose-detecting) data, including ragged sizes that exercise every bounds guard."""
import pytest
import torch
import torch.nn.functional as F

from msclip_amd import hip, packing as P

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32, dev="cuda"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)


def close(got, ref, atol, rtol=0.0):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.4g} (ref absmax {ref.abs().max().item():.4g})"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 2304, 768), (1000, 768, 3072), (77, 512, 768), (513, 48, 128)])
def test_gemm_dense_bias_and_shapes(gpu_device, M, N, K):
    x, w, b = rnd(M, K, seed=1, dtype=BF), rnd(N, K, seed=2, scale=0.05, dtype=BF), rnd(N, seed=3)
    ref = x.float() @ w.float().t() + b
    out = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
    hip.gemm(x, w, out, bias=b)
    close(out, ref, 2e-2, 1e-2)
    outf = torch.empty(M, N, dtype=torch.float32, device="cuda")
    hip.gemm(x, w, outf, bias=b, alpha=0.5)
    close(outf, 0.5 * (x.float() @ w.float().t()) + b, 2e-3, 1e-4)


@pytest.mark.parametrize("tile", [1, 4])
@pytest.mark.parametrize("M,N,K", [(1000, 768, 3072), (700, 520, 128), (256, 256, 64), (2051, 1096, 768)])
def test_gemm_every_tile_config(gpu_device, tile, M, N, K):
    """The dense main loops (128x128 two-buffer, 256x256 ping-pong) on ragged edges, all three epilogue kinds."""
    x, w, b = rnd(M, K, seed=11, dtype=BF), rnd(N, K, seed=12, scale=0.05, dtype=BF), rnd(N, seed=13)
    base = x.float() @ w.float().t() + b
    out = torch.full((M + 3, N), float("nan"), dtype=BF, device="cuda")
    hip.gemm(x, w, out[:M], bias=b, act=hip.ACT_QUICKGELU, tile=tile)
    close(out[:M], base * torch.sigmoid(1.702 * base), 2e-2, 1e-2)
    assert bool(torch.isnan(out[M:].float()).all())                              # nothing written past M
    r32 = rnd(M, N, seed=14)
    xres = r32.clone()
    for _ in range(2):                                                           # back-to-back launches: ring state, races
        xres.copy_(r32)
        hip.gemm(x, w, xres, bias=b, resid=xres, resid_kind=hip.RESID_F32, tile=tile)
        close(xres, r32 + base, 4e-3, 1e-4)


@pytest.mark.parametrize("tile", [4])
@pytest.mark.parametrize("nt_m,nt_n", [(1, 1), (3, 2), (5, 3), (2, 4), (7, 5), (4, 6), (3, 7), (9, 9), (40, 3), (300, 1), (5, 17), (70, 9), (3, 24)])
def test_gemm_pingpong_tile_map_covers_every_tile(gpu_device, tile, nt_m, nt_n):
    """The ping-pong kernels' tile-id -> origin maps (XCD remap, column groups of four 256-column / eight 128-column tiles,
    reciprocal-multiply divisions): every tile is computed exactly once for full, ragged and single-column group layouts,
    with fewer and with more tiles than workgroups."""
    tn = 256 if tile == 4 else 128
    M, N, K = nt_m * 256 - 19, nt_n * tn - 40, 64
    x, w = rnd(M, K, seed=21, dtype=BF), rnd(N, K, seed=22, scale=0.1, dtype=BF)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    acc = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    hip.gemm(x, w, out, tile=tile)
    ref = x.float() @ w.float().t()
    close(out, ref, 2e-3, 1e-3)
    hip.gemm(x, w, acc, resid=acc, resid_kind=hip.RESID_F32, tile=tile)              # a tile done twice would add twice
    close(acc, ref, 2e-3, 1e-3)


@pytest.mark.parametrize("M,N,K,ldx", [(5000, 96, 64, 48), (4099, 48, 64, 48), (8000, 192, 128, 96), (4700, 192, 64, 64),
                                         (4100, 768, 192, 192), (6000, 384, 192, 192), (4096, 40, 64, 64)])
def test_gemm_streaming_small_k(gpu_device, M, N, K, ldx):
    """Pointwise-conv shapes (K <= 192, rows of ldx < K elements overlapping the next row against zero weights)."""
    cin = ldx
    x = rnd(M + 2, ldx, seed=21, dtype=BF)                                       # 2 rows of slack past M
    w = torch.zeros(N, K, dtype=BF, device="cuda")
    w[:, :cin] = rnd(N, cin, seed=22, scale=0.08, dtype=BF)
    b = rnd(N, seed=23)
    base = x[:M].float() @ w[:, :cin].float().t() + b
    assert hip.gemm_variant(hip.describe_gemm(0, M, N, K, ldx=ldx)) == "stream"
    out = torch.full((M + 1, N), float("nan"), dtype=BF, device="cuda")
    hip.gemm(x, w, out[:M], M=M, bias=b, act=hip.ACT_RELU, ldx=ldx)
    close(out[:M], F.relu(base), 2e-2, 1e-2)
    assert bool(torch.isnan(out[M].float()).all())
    r16 = rnd(M, N, seed=24, dtype=BF)
    hip.gemm(x, w, out[:M], M=M, bias=b, act=hip.ACT_RELU, resid=r16, resid_kind=hip.RESID_BF16, ldx=ldx)
    close(out[:M], F.relu(base + r16.float()), 3e-2, 1e-2)
    o32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    hip.gemm(x, w, o32, M=M, bias=b, alpha=0.5, ldx=ldx)
    close(o32, 0.5 * (base - b) + b, 2e-3, 1e-4)
    ref1 = out[:M].clone()
    hip.gemm(x, w, out[:M], M=M, bias=b, act=hip.ACT_RELU, resid=r16, resid_kind=hip.RESID_BF16, ldx=ldx, tile=1)
    close(out[:M], ref1, 2e-2, 1e-2)                                             # agrees with the 128x128 kernel


@pytest.mark.parametrize("tile", [4])
def test_gemm_pingpong_race_screen(gpu_device, tile):
    """The ping-pong kernels order LDS-DMA writes, fragment reads and ring-slot reuse by counted waits and barriers only;
    a misplaced one shows up as rare wrong tiles.  Many launches, every output element, bitwise-equal results."""
    M, N, K = 8192 + 77, 1024, 1536                                            # 33 x 4 tiles, ragged last row block
    x, w, b = rnd(M, K, seed=51, dtype=BF), rnd(N, K, seed=52, scale=0.04, dtype=BF), rnd(N, seed=53)
    ref = x.float() @ w.float().t() + b
    first = None
    out = torch.empty(M, N, dtype=BF, device="cuda")
    for it in range(25):
        out.fill_(float("nan"))
        hip.gemm(x, w, out, bias=b, tile=tile)
        if first is None:
            close(out, ref, 2e-2, 1e-2)
            first = out.clone()
        else:
            assert torch.equal(out, first), f"launch {it} differs from launch 0"
    r32 = rnd(M, N, seed=54)
    acc = r32.clone()
    for it in range(10):
        acc.copy_(r32)
        hip.gemm(x, w, acc, bias=b, resid=acc, resid_kind=hip.RESID_F32, tile=tile)
        if it == 0:
            close(acc, r32 + ref, 4e-3, 1e-4)
            first = acc.clone()
        else:
            assert torch.equal(acc, first), f"launch {it} differs from launch 0"


@pytest.mark.parametrize("tile", [4])
@pytest.mark.parametrize("act", [hip.ACT_NONE, hip.ACT_QUICKGELU])
def test_gemm_pingpong_race_screen_multi_tile(gpu_device, act, tile):
    """Same screen with several tiles per workgroup (768 tiles on <= 256 workgroups, 12 K-tiles each): the DMA stream
    crossing tile boundaries under the epilogue, and the first wait of a tile that leaves the previous tile's stores
    in flight, both epilogue families."""
    M, N, K = 256 * 64 - 5, 256 * 12, 768
    x, w, b = rnd(M, K, seed=55, dtype=BF), rnd(N, K, seed=56, scale=0.04, dtype=BF), rnd(N, seed=57)
    ref = x.float() @ w.float().t() + b
    if act == hip.ACT_QUICKGELU:
        ref = ref * torch.sigmoid(1.702 * ref)
    out = torch.empty(M, N, dtype=BF, device="cuda")
    first = None
    for it in range(12):
        out.fill_(float("nan"))
        hip.gemm(x, w, out, bias=b, act=act, tile=tile)
        if first is None:
            close(out, ref, 2e-2, 1e-2)
            first = out.clone()
        else:
            assert torch.equal(out, first), f"launch {it} differs from launch 0"
    del out, first
    if act == hip.ACT_NONE:
        r32 = rnd(M, N, seed=58)
        acc = r32.clone()
        for it in range(6):
            acc.copy_(r32)
            hip.gemm(x, w, acc, bias=b, resid=acc, resid_kind=hip.RESID_F32, tile=tile)
            if it == 0:
                close(acc, r32 + ref, 4e-3, 1e-4)
                first = acc.clone()
            else:
                assert torch.equal(acc, first), f"launch {it} differs from launch 0"


def test_gemm_epilogues(gpu_device):
    M, N, K = 333, 256, 192
    x, w, b = rnd(M, K, seed=4, dtype=BF), rnd(N, K, seed=5, scale=0.08, dtype=BF), rnd(N, seed=6)
    base = x.float() @ w.float().t() + b
    out = torch.empty(M, N, dtype=BF, device="cuda")
    hip.gemm(x, w, out, bias=b, act=hip.ACT_QUICKGELU)
    close(out, base * torch.sigmoid(1.702 * base), 2e-2, 1e-2)
    hip.gemm(x, w, out, bias=b, act=hip.ACT_RELU)
    close(out, F.relu(base), 2e-2, 1e-2)
    r32 = rnd(M, N, seed=7)
    xres = r32.clone()
    hip.gemm(x, w, xres, bias=b, resid=xres, resid_kind=hip.RESID_F32)          # in-place residual update
    close(xres, r32 + base, 2e-3, 1e-4)
    r16 = rnd(M, N, seed=8, dtype=BF)
    hip.gemm(x, w, out, bias=b, resid=r16, resid_kind=hip.RESID_BF16, act=hip.ACT_RELU)
    close(out, F.relu(base + r16.float()), 3e-2, 1e-2)


def test_gemm_ragged_n_tail_path(gpu_device):
    """N and ld not multiples of 4 (ragged global batch in the logits GEMM)."""
    M, N, K = 37, 5, 128
    x, w, b = rnd(M, K, seed=41, dtype=BF), rnd(N, K, seed=42, scale=0.1, dtype=BF), rnd(N, seed=43)
    out = torch.full((M + 1, N), 3.0, dtype=torch.float32, device="cuda")
    hip.gemm(x, w, out[:M], bias=b, alpha=2.0)
    close(out[:M], 2.0 * (x.float() @ w.float().t()) + b, 2e-3, 1e-4)
    assert bool((out[M] == 3.0).all())
    o16 = torch.empty(M, N, dtype=BF, device="cuda")
    hip.gemm(x, w, o16, bias=b, act=hip.ACT_RELU)
    close(o16, F.relu(x.float() @ w.float().t() + b), 2e-2, 1e-2)


def test_gemm_token_scatter_with_table(gpu_device):
    """Stem last_conv epilogue: + positional row (p + 1), scatter to token row b*L + 1 + p."""
    B, g2, D = 3, 49, 256
    L = g2 + 1
    x, w = rnd(B * g2, D, seed=9, dtype=BF), rnd(D, D, seed=10, scale=0.05, dtype=BF)
    pos = rnd(L, D, seed=11)
    X = torch.zeros(B * L + 7, D, dtype=torch.float32, device="cuda")
    hip.gemm(x, w, X, M=B * g2, resid=pos, resid_kind=hip.RESID_TABLE, rpg=g2, radd=1, roff=1)
    ref = (x.float() @ w.float().t()).reshape(B, g2, D) + pos[1:]
    got = X[:B * L].reshape(B, L, D)
    close(got[:, 1:], ref, 2e-3, 1e-4)
    assert float(got[:, 0].abs().max()) == 0.0 and float(X[B * L:].abs().max()) == 0.0


@pytest.mark.parametrize("tile", [0, 1, 4])
def test_gemm_token_scatter_every_tile_config(gpu_device, tile):
    """The stem -> token-row scatter with the positional table at a batch where the large-tile kernels take it."""
    B, g2, D, K = 700, 49, 768, 768                                # M = 34300: 134 row tiles of 256
    x, w = rnd(B * g2, K, seed=71, dtype=BF), rnd(D, K, seed=72, scale=0.05, dtype=BF)
    pos = rnd(g2 + 1, D, seed=73)
    X = torch.full((B * (g2 + 1), D), float("nan"), dtype=torch.float32, device="cuda")
    hip.gemm(x, w, X, M=B * g2, resid=pos, resid_kind=hip.RESID_TABLE, rpg=g2, radd=1, roff=1, tile=tile)
    ref = (x.float() @ w.float().t()).reshape(B, g2, D) + pos[1:]
    got = X.reshape(B, g2 + 1, D)
    close(got[:, 1:], ref, 4e-3, 1e-4)
    assert bool(torch.isnan(got[:, 0]).all())                                   # class-token rows untouched


def test_gemm_strided_operands_for_logits(gpu_device):
    """Both operands are column slices of the packed [N, 2, E] feature buffer."""
    n, E = 200, 512
    packed = F.normalize(rnd(n, 2, E, seed=12), dim=-1).to(BF)
    out = torch.empty(n, n, dtype=torch.float32, device="cuda")
    hip.gemm(packed[:, 0], packed[:, 1], out, alpha=14.0)
    close(out, 14.0 * packed[:, 0].float() @ packed[:, 1].float().t(), 2e-3)


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride,pad", [
    (2, 28, 48, 96, 3, 2, 1), (3, 14, 96, 48, 1, 1, 0), (2, 14, 192, 384, 3, 1, 1), (2, 16, 48, 96, 1, 2, 0),
    (1, 7, 768, 768, 1, 1, 0), (2, 56, 48, 48, 3, 2, 1),
    # M >= 4096: the streaming kernel (weights resident in LDS) takes these
    (3, 112, 48, 96, 3, 2, 1), (2, 112, 48, 48, 3, 2, 1), (5, 56, 48, 96, 1, 2, 0), (6, 56, 96, 192, 1, 2, 0),
    (24, 28, 192, 384, 1, 2, 0), (3, 90, 48, 40, 3, 2, 1),
    # N % 192 == 0 and enough tiles: the 256 x 192 two-buffer configuration
    (45, 56, 96, 192, 3, 2, 1), (200, 14, 384, 768, 3, 2, 1),
    # input channels a multiple of 64 and >= 128 tiles of 256 x 256: the ping-pong kernel in implicit-conv mode
    (180, 28, 192, 384, 3, 2, 1), (170, 28, 192, 192, 3, 2, 1), (700, 14, 384, 768, 3, 2, 1), (150, 28, 64, 200, 3, 1, 1)])
def test_gemm_implicit_conv(gpu_device, B, H, Cin, Cout, k, stride, pad):
    x = rnd(B, H, H, Cin, seed=13, dtype=BF)                      # NHWC
    w = rnd(Cout, Cin, k, k, seed=14, scale=(2.0 / (Cin * k * k)) ** 0.5)
    b = rnd(Cout, seed=15)
    spec = P.ConvSpec(w, b, H, H, stride, pad).to("cuda")
    out = torch.full((B * spec.h_out * spec.w_out, Cout), float("nan"), dtype=BF, device="cuda")
    hip.gemm(x, spec.weight, out, M=out.shape[0], N=Cout, bias=spec.bias, act=hip.ACT_RELU, conv=spec.geometry(),
             ktab=spec.ktab)
    ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.to(BF).float(), b, stride=stride, padding=pad))
    close(out.reshape(B, spec.h_out, spec.w_out, Cout), ref.permute(0, 2, 3, 1), 3e-2, 1e-2)


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride,pad", [(2, 14, 64, 96, 3, 2, 1), (3, 9, 128, 264, 3, 1, 1), (1, 20, 192, 40, 1, 1, 0)])
def test_gemm_conv_pingpong_forced_small(gpu_device, B, H, Cin, Cout, k, stride, pad):
    """The ping-pong kernel's implicit-conv mode on a single ragged tile (rows past M, columns past N, every border)."""
    x = rnd(B, H, H, Cin, seed=81, dtype=BF)
    w = rnd(Cout, Cin, k, k, seed=82, scale=(2.0 / (Cin * k * k)) ** 0.5)
    b = rnd(Cout, seed=83)
    spec = P.ConvSpec(w, b, H, H, stride, pad).to("cuda")
    M = B * spec.h_out * spec.w_out
    out = torch.full((M + 2, Cout), float("nan"), dtype=BF, device="cuda")
    hip.gemm(x, spec.weight, out[:M], M=M, N=Cout, bias=spec.bias, act=hip.ACT_RELU, conv=spec.geometry(),
             ktab=spec.ktab, tile=4)
    ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.to(BF).float(), b, stride=stride, padding=pad))
    close(out[:M].reshape(B, spec.h_out, spec.w_out, Cout), ref.permute(0, 2, 3, 1), 3e-2, 1e-2)
    assert bool(torch.isnan(out[M:].float()).all())


@pytest.mark.parametrize("B,L,causal", [(3, 50, False), (5, 77, True), (2, 197, False), (1, 1, False), (2, 33, True),
                                        (2, 64, True), (1, 96, False), (2, 257, False), (1, 288, True), (2, 230, False)])
def test_attention(gpu_device, B, L, causal):
    Hh, D = 12, 768
    qkv = rnd(B * L + 5, 3 * D, seed=16, dtype=BF)                 # extra rows: nothing may read/write past B*L
    out = torch.full((B * L + 5, D), 7.0, dtype=BF, device="cuda")
    hip.attention(qkv[:B * L], out[:B * L], B, L, Hh, causal)
    q, k, v = qkv[:B * L].float().reshape(B, L, 3, Hh, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf"), device="cuda").triu_(1)
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, D)
    close(out[:B * L], ref, 2e-2, 2e-2)
    assert bool((out[B * L:] == 7.0).all())


@pytest.mark.parametrize("B,L,causal", [(5, 50, False), (6, 77, True), (3, 197, False), (2, 1, True), (3, 64, True), (2, 130, True),
                                        (3, 257, False), (2, 288, True)])
def test_attention_single_query_per_sample(gpu_device, B, L, causal):
    """msclip_attention_lastq: the class row (all keys) / a caption's EOT row (keys up to itself) only.  Against fp32 softmax
    attention of that one query, and against the same rows of the full kernel.  The q columns of the token matrix are
    poisoned: only k | v may be read from it."""
    Hh, D, base = 12, 768, 7
    qkv = rnd(base + B * L + 3, 3 * D, seed=18, dtype=BF)
    g = torch.Generator().manual_seed(5)
    pos = torch.randint(0, L, (B,), generator=g) if causal else torch.zeros(B, dtype=torch.long)
    rows = (base + torch.arange(B) * L + pos).to(torch.int32).cuda()
    q = qkv[rows.long(), :D].contiguous()
    full = torch.empty(B * L, D, dtype=BF, device="cuda")
    if L <= 288:
        hip.attention(qkv[base:base + B * L], full, B, L, Hh, causal)
    poisoned = qkv.clone()
    poisoned[:, :D] = float("nan")
    out = torch.full((B + 2, D), 7.0, dtype=BF, device="cuda")
    hip.attention_lastq(q, poisoned, out, B, L, Hh, last_row=rows if causal else None, row_base=base)
    for b in range(B):
        nk = int(pos[b]) + 1 if causal else L
        blk = qkv[base + b * L: base + b * L + nk].float()
        k, v = blk[:, D:2 * D].reshape(nk, Hh, 64), blk[:, 2 * D:].reshape(nk, Hh, 64)
        s = torch.einsum("hd,khd->hk", q[b].float().reshape(Hh, 64), k)
        ref = torch.einsum("hk,khd->hd", torch.softmax(s, -1), v).reshape(D)
        close(out[b], ref, 2e-2, 2e-2)
        if L <= 288:
            close(out[b], full[b * L + int(pos[b])].float(), 2e-2, 1e-2)
    assert bool((out[B:] == 7.0).all())
    assert hip.lib().msclip_attention_lastq(None, D, None, 3 * D, None, D, 1, 8, Hh, None, 0, None) == -1
    assert hip.lib().msclip_attention_lastq(q.data_ptr(), D, qkv.data_ptr(), 3 * D, out.data_ptr(), D, 1, 300, Hh, None, 0, None) == -1


def test_attention_forced_sharp_softmax(gpu_device):
    """One key dominates each query (exercises max-subtraction across the half-wave exchange)."""
    B, L, Hh, D = 2, 77, 12, 768
    qkv = rnd(B * L, 3 * D, seed=17, dtype=BF)
    qkv[:, :D] *= 6.0
    out = torch.empty(B * L, D, dtype=BF, device="cuda")
    hip.attention(qkv, out, B, L, Hh, True)
    q, k, v = qkv.float().reshape(B, L, 3, Hh, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) + torch.full((L, L), float("-inf"), device="cuda").triu_(1)
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, D)
    close(out, ref, 3e-2, 2e-2)


def test_layernorm_variants(gpu_device):
    M, C = 301, 768
    x = rnd(M, C, seed=18, scale=3.0) + 0.7
    g, b = rnd(C, seed=19) + 1.0, rnd(C, seed=20)

    def ref_ln(t):
        u = t.mean(-1, keepdim=True)
        s = (t - u).pow(2).mean(-1, keepdim=True)
        return g * ((t - u) / torch.sqrt(s + 1e-12)) + b
    o32 = torch.empty(M, C, dtype=torch.float32, device="cuda")
    hip.layernorm(x, g, b, o32, M)
    close(o32, ref_ln(x), 1e-4, 1e-5)
    o16 = torch.empty(M, C, dtype=BF, device="cuda")
    raw = torch.zeros(M, C, dtype=torch.float32, device="cuda")
    hip.layernorm(x, g, b, o16, M, raw_out=raw)
    close(o16, ref_ln(x), 2e-2, 1e-2)
    assert torch.equal(raw, x)
    idx = torch.tensor([5, 300, 0, 17], dtype=torch.int32, device="cuda")
    o = torch.empty(4, C, dtype=torch.float32, device="cuda")
    hip.layernorm(x, g, b, o, 4, row_idx=idx)
    close(o, ref_ln(x[idx.long()]), 1e-4, 1e-5)
    o = torch.empty(6, C, dtype=torch.float32, device="cuda")
    hip.layernorm(x, g, b, o, 6, row_mul=50)
    close(o, ref_ln(x[::50][:6]), 1e-4, 1e-5)
    xin = x.clone()
    hip.layernorm(xin, g, b, xin, M)                               # in place
    close(xin, ref_ln(x), 1e-4, 1e-5)
    x512 = rnd(9, 512, seed=21)
    o = torch.empty(9, 512, dtype=torch.float32, device="cuda")
    hip.layernorm(x512, g[:512].contiguous(), b[:512].contiguous(), o, 9)
    u = x512.mean(-1, keepdim=True)
    close(o, g[:512] * ((x512 - u) / torch.sqrt((x512 - u).pow(2).mean(-1, keepdim=True) + 1e-12)) + b[:512], 1e-4, 1e-5)


@pytest.mark.parametrize("M,split,C", [(4097, 2049, 768), (5000, 5000, 768), (4100, 0, 512), (8192, 4096, 256)])
def test_layernorm_pair_kernel(gpu_device, monkeypatch, M, split, C):
    """The two-rows-per-wave LayerNorm of the big launches: odd row counts, an odd / even / absent modality boundary,
    fp32 and bf16 outputs, in place; agrees with the wave-per-row kernel to fp32 rounding (the compiler contracts the
    affine step differently)."""
    x = rnd(M, C, seed=71, scale=3.0) + 0.5
    g1, b1, g2, b2 = rnd(C, seed=72) * 0.2 + 1, rnd(C, seed=73) * 0.1, rnd(C, seed=74) * 0.2 + 1, rnd(C, seed=75) * 0.1

    def ln(v, g, b):
        u = v.mean(-1, keepdim=True)
        s = ((v - u) ** 2).mean(-1, keepdim=True)
        return g * ((v - u) / torch.sqrt(s + 1e-12)) + b
    ref = torch.cat([ln(x[:split], g1, b1), ln(x[split:], g2, b2)])
    out = torch.full((M + 1, C), float("nan"), dtype=BF, device="cuda")
    hip.layernorm_split(x, g1, b1, g2, b2, split, out[:M], M)
    close(out[:M], ref, 2e-2, 1e-2)
    assert bool(torch.isnan(out[M:].float()).all())
    o32 = torch.empty(M, C, dtype=torch.float32, device="cuda")
    hip.layernorm_split(x, g1, b1, g2, b2, split, o32, M)
    close(o32, ref, 1e-4, 1e-5)
    monkeypatch.setenv("MSCLIP_LN_SINGLE_ROW", "1")
    one = torch.empty(M, C, dtype=torch.float32, device="cuda")
    hip.layernorm_split(x, g1, b1, g2, b2, split, one, M)
    close(o32, one, 2e-6, 2e-6)
    monkeypatch.setenv("MSCLIP_LN_SINGLE_ROW", "0")
    xin = x.clone()
    hip.layernorm(xin, g1, b1, xin, M)                                           # in place, single parameter set
    close(xin, ln(x, g1, b1), 1e-4, 1e-5)


def test_layernorm_split_parameters(gpu_device):
    M, C, split = 700, 768, 257
    x = rnd(M, C, seed=61, scale=3.0) + 0.5
    g1, b1, g2, b2 = rnd(C, seed=62) * 0.2 + 1, rnd(C, seed=63) * 0.1, rnd(C, seed=64) * 0.2 + 1, rnd(C, seed=65) * 0.1
    out = torch.empty(M, C, dtype=BF, device="cuda")
    hip.layernorm_split(x, g1, b1, g2, b2, split, out, M)

    def ln(v, g, b):
        u = v.mean(-1, keepdim=True)
        s = ((v - u) ** 2).mean(-1, keepdim=True)
        return g * ((v - u) / torch.sqrt(s + 1e-12)) + b
    close(out[:split], ln(x[:split], g1, b1), 2e-2, 1e-2)
    close(out[split:], ln(x[split:], g2, b2), 2e-2, 1e-2)
    ref1 = torch.empty(split, C, dtype=BF, device="cuda")
    hip.layernorm(x[:split], g1, b1, ref1, split)
    assert torch.equal(out[:split], ref1)                                       # bitwise the single-set kernel


def test_embed_tokens_and_eot(gpu_device):
    B, L, C, V = 5, 77, 768, 1000
    emb, pos = rnd(V, C, seed=22), rnd(L, C, seed=23)
    g = torch.Generator().manual_seed(24)
    tok = torch.randint(1, V - 2, (B, L), generator=g)
    eot_pos = [3, 76, 10, 1, 40]
    for i, p in enumerate(eot_pos):
        tok[i, p] = V - 1
        tok[i, p + 1:] = 0
    tok[2, 20] = V - 1                                             # duplicate maximum: argmax takes the first
    tok = tok.cuda()
    X = torch.zeros(11 + B * L, C, dtype=torch.float32, device="cuda")
    eot = torch.empty(B, dtype=torch.int32, device="cuda")
    hip.embed_tokens(tok, emb, pos, X, eot, 11)
    close(X[11:].reshape(B, L, C), emb[tok] + pos, 1e-6)
    assert float(X[:11].abs().max()) == 0.0
    assert eot.tolist() == [11 + i * L + int(tok[i].argmax()) for i in range(B)]


def test_fill_cls_adapter_l2norm(gpu_device):
    B, g_, C = 3, 7, 768
    L = g_ * g_ + 1
    cls, pos = rnd(C, seed=25), rnd(L, C, seed=26)
    X = torch.zeros(B * L, C, dtype=torch.float32, device="cuda")
    hip.fill_cls(cls, pos, X, B, L)
    close(X.reshape(B, L, C)[:, 0], (cls + pos[0]).expand(B, C), 1e-6)
    assert float(X.reshape(B, L, C)[:, 1:].abs().max()) == 0.0

    x = rnd(B * L, C, seed=27)
    t = rnd(B * g_ * g_, C, seed=28)
    dww, dwb = rnd(9, C, seed=29, scale=0.3), rnd(C, seed=30, scale=0.1)
    ga, be = rnd(C, seed=31) + 1.0, rnd(C, seed=32)
    out = torch.empty(B * L, C, dtype=torch.float32, device="cuda")
    hip.adapter_combine_ln(x, t, dww, dwb, ga, be, out, B, L, g_, True)
    xb = x.reshape(B, L, C)
    grid = xb[:, 1:].transpose(1, 2).reshape(B, C, g_, g_)
    bo = F.conv2d(grid, dww.t().reshape(C, 1, 3, 3), dwb, padding=1, groups=C).flatten(2).transpose(1, 2)
    v = torch.cat([2 * xb[:, :1], bo + t.reshape(B, g_ * g_, C)], 1)
    u = v.mean(-1, keepdim=True)
    ref = ga * ((v - u) / torch.sqrt((v - u).pow(2).mean(-1, keepdim=True) + 1e-12)) + be
    close(out.reshape(B, L, C), ref, 2e-4, 1e-5)

    f = rnd(37, 512, seed=33, scale=4.0)
    o32 = torch.empty(37, 512, dtype=torch.float32, device="cuda")
    packed = torch.zeros(37, 2, 512, dtype=BF, device="cuda")
    hip.l2norm(f, o32, packed[:, 1])
    close(o32, f / f.norm(dim=-1, keepdim=True), 1e-6, 1e-6)
    close(packed[:, 1], f / f.norm(dim=-1, keepdim=True), 1e-3)
    assert float(packed[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("kernel", ["sample", "gridrow", "per_token"])
@pytest.mark.parametrize("B,g_,C,usecls", [(5, 7, 768, True), (3, 14, 768, False), (2, 1, 768, True), (9, 3, 256, True)])
def test_adapter_combine_ln(gpu_device, monkeypatch, B, g_, C, usecls, kernel):
    """Lateral adapter bottom half + sum + LayerNorm (M.py:1763-1777): the workgroup-per-sample kernel (filters in LDS: the
    default), the wave-per-grid-row kernel (filters in registers) and the wave-per-token one, grids 1 / 3 / 7 / 14, both
    class-token modes."""
    monkeypatch.setenv("MSCLIP_ADAPTER_PER_TOKEN", "1" if kernel == "per_token" else "0")
    monkeypatch.setenv("MSCLIP_ADAPTER_SAMPLE", "1" if kernel == "sample" else "0")
    L = g_ * g_ + 1
    x, t = rnd(B * L, C, seed=61), rnd(B * g_ * g_, C, seed=62)
    dww, dwb = rnd(9, C, seed=63, scale=0.3), rnd(C, seed=64, scale=0.1)
    ga, be = 1.0 + rnd(C, seed=65, scale=0.1), rnd(C, seed=66, scale=0.1)
    out = torch.full((B * L + 1, C), float("nan"), dtype=torch.float32, device="cuda")
    hip.adapter_combine_ln(x, t, dww, dwb, ga, be, out[:B * L], B, L, g_, usecls)
    xb = x.reshape(B, L, C)
    grid = xb[:, 1:].transpose(1, 2).reshape(B, C, g_, g_)
    bo = F.conv2d(grid, dww.t().reshape(C, 1, 3, 3), dwb, padding=1, groups=C).flatten(2).transpose(1, 2)
    v = torch.cat([(2 if usecls else 1) * xb[:, :1], bo + t.reshape(B, g_ * g_, C)], 1)
    u = v.mean(-1, keepdim=True)
    ref = ga * ((v - u) / torch.sqrt((v - u).pow(2).mean(-1, keepdim=True) + 1e-12)) + be
    close(out[:B * L].reshape(B, L, C), ref, 2e-4, 1e-5)
    assert bool(torch.isnan(out[B * L:]).all())


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_stem_dual_conv(gpu_device, dtype):
    B, S = 3, 64
    img = rnd(B, 3, S, S, seed=34, dtype=dtype)
    w, b = rnd(27, 96, seed=35, scale=0.3), rnd(96, seed=36, scale=0.2)
    oa = torch.empty(B * 32 * 32, 48, dtype=BF, device="cuda")
    ob = torch.empty(B * 32 * 32, 48, dtype=BF, device="cuda")
    hip.stem_conv_dual(img, w, b, oa, ob)
    ref = F.relu(F.conv2d(img.float(), w.t().reshape(96, 3, 3, 3), b, stride=2, padding=1)).permute(0, 2, 3, 1)
    close(oa.reshape(B, 32, 32, 48), ref[..., :48], 2e-2, 1e-2)
    close(ob.reshape(B, 32, 32, 48), ref[..., 48:], 2e-2, 1e-2)


@pytest.mark.parametrize("form", ["8wave", "4wave"])
@pytest.mark.parametrize("B,H,Cout", [(2, 32, 48), (3, 23, 48), (1, 112, 48), (2, 20, 96), (600, 8, 48)])
def test_fused_conv1x1_conv3x3s2(gpu_device, monkeypatch, B, H, Cout, form):
    """Bottleneck conv1 -> conv2 without the intermediate map: tile interiors, halos, odd sizes, many tiles per CU
    (more tiles than workgroups: the persistent loop, both window buffers, the prefetch chain)."""
    monkeypatch.setenv("MSCLIP_FRONT_4WAVE", "1" if form == "4wave" else "0")
    x = rnd(B, H, H, 48, seed=91, dtype=BF)
    w1, b1 = rnd(48, 48, 1, 1, seed=92, scale=(2.0 / 48) ** 0.5), rnd(48, seed=93, scale=0.2)
    w2, b2 = rnd(Cout, 48, 3, 3, seed=94, scale=(2.0 / 432) ** 0.5), rnd(Cout, seed=95, scale=0.2)
    c1 = P.ConvSpec(w1, b1, H, H, 1, 0).to("cuda")
    c2 = P.ConvSpec(w2, b2, H, H, 2, 1).to("cuda")
    Ho = c2.h_out
    out = torch.full((B * Ho * Ho + 3, Cout), float("nan"), dtype=BF, device="cuda")
    hip.conv1x1_conv3x3s2(x, c1.weight, c1.bias, c2.weight, c2.bias, out, B, H, H)
    xn = x.float().permute(0, 3, 1, 2)
    t1 = F.relu(F.conv2d(xn, w1.to(BF).float(), b1)).to(BF).float()
    ref = F.relu(F.conv2d(t1, w2.to(BF).float(), b2, stride=2, padding=1)).permute(0, 2, 3, 1)
    close(out[:B * Ho * Ho].reshape(B, Ho, Ho, Cout), ref, 3e-2, 1e-2)
    assert bool(torch.isnan(out[B * Ho * Ho:].float()).all())


@pytest.mark.parametrize("B,H", [(2, 32), (3, 23), (1, 112), (700, 8)])
def test_fused_convresblock48_s2(gpu_device, B, H):
    """The whole stride-2 bottleneck (conv1, conv2, conv3 + strided shortcut + add, ReLUs) in one launch against the
    unfused chain's arithmetic: intermediates rounded to bf16 where the chain stores them."""
    x = rnd(B, H, H, 48, seed=91, dtype=BF)
    w1, b1 = rnd(48, 48, 1, 1, seed=92, scale=(2.0 / 48) ** 0.5), rnd(48, seed=93, scale=0.2)
    w2, b2 = rnd(48, 48, 3, 3, seed=94, scale=(2.0 / 432) ** 0.5), rnd(48, seed=95, scale=0.2)
    w3, b3 = rnd(96, 48, 1, 1, seed=96, scale=(1.0 / 48) ** 0.5), rnd(96, seed=97, scale=0.2)
    wr, br = rnd(96, 48, 1, 1, seed=98, scale=(1.0 / 48) ** 0.5), rnd(96, seed=99, scale=0.2)
    c1 = P.ConvSpec(w1, b1, H, H, 1, 0).to("cuda")
    c2 = P.ConvSpec(w2, b2, H, H, 2, 1).to("cuda")
    cr = P.ConvSpec(wr, br, H, H, 2, 0).to("cuda")
    c3 = P.ConvSpec(w3, b3, c2.h_out, c2.w_out, 1, 0).to("cuda")
    Ho = c2.h_out
    assert cr.h_out == Ho
    out = torch.full((B * Ho * Ho + 3, 96), float("nan"), dtype=BF, device="cuda")
    hip.convresblock48_s2(x, c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, cr.weight, (c3.bias + cr.bias).contiguous(),
                          out, B, H, H)
    xn = x.float().permute(0, 3, 1, 2)
    t1 = F.relu(F.conv2d(xn, w1.to(BF).float(), b1)).to(BF).float()
    t2 = F.relu(F.conv2d(t1, w2.to(BF).float(), b2, stride=2, padding=1)).to(BF).float()
    tr = F.conv2d(xn, wr.to(BF).float(), br, stride=2)
    ref = F.relu(F.conv2d(t2, w3.to(BF).float(), b3) + tr).permute(0, 2, 3, 1)
    close(out[:B * Ho * Ho].reshape(B, Ho, Ho, 96), ref, 4e-2, 1e-2)
    assert bool(torch.isnan(out[B * Ho * Ho:].float()).all())


@pytest.mark.parametrize("form", ["8wave", "4wave"])
@pytest.mark.parametrize("dtype,B,S,Cout", [(torch.float32, 3, 64, 96), (BF, 2, 48, 96), (torch.float32, 2, 224, 96),
                                            (torch.float32, 2, 40, 48), (torch.float32, 300, 32, 96), (BF, 2, 42, 96)])
def test_stem_dual_fused_with_next_conv(gpu_device, monkeypatch, dtype, B, S, Cout, form):
    """Stem conv1 + parallel stage 0 + the stem's 3x3/s2 stage 0 in one pass: branch b as the unfused kernel writes
    it, the 96-channel map as the unfused chain computes it.  300 x 32 x 32 gives every workgroup several tiles (the
    two-deep prefetch and both window buffers); width 42 is not a multiple of 4 (single-group kernel either way)."""
    monkeypatch.setenv("MSCLIP_FRONT_4WAVE", "1" if form == "4wave" else "0")
    img = rnd(B, 3, S, S, seed=34, dtype=dtype)
    w, b = rnd(27, 96, seed=35, scale=0.3), rnd(96, seed=36, scale=0.2)
    w2, b2 = rnd(Cout, 48, 3, 3, seed=96, scale=(2.0 / 432) ** 0.5), rnd(Cout, seed=97, scale=0.2)
    Hm = S // 2
    c2 = P.ConvSpec(w2, b2, Hm, Hm, 2, 1).to("cuda")
    Ho = c2.h_out
    ob = torch.full((B * Hm * Hm + 2, 48), float("nan"), dtype=BF, device="cuda")
    out = torch.full((B * Ho * Ho + 2, Cout), float("nan"), dtype=BF, device="cuda")
    hip.stem_dual_conv3x3s2(img, w, b, ob, c2.weight, c2.bias, out)
    # operands rounded to bf16 as the kernel rounds them: what is left is summation order and the output rounding
    mid = F.relu(F.conv2d(img.to(BF).float(), w.to(BF).float().t().reshape(96, 3, 3, 3), b, stride=2, padding=1))
    close(ob[:B * Hm * Hm].reshape(B, Hm, Hm, 48), mid[:, 48:].permute(0, 2, 3, 1), 2e-3, 8e-3)
    ref = F.relu(F.conv2d(mid[:, :48].to(BF).float(), w2.to(BF).float(), b2, stride=2, padding=1)).permute(0, 2, 3, 1)
    close(out[:B * Ho * Ho].reshape(B, Ho, Ho, Cout), ref, 4e-2, 1e-2)
    assert bool(torch.isnan(ob[B * Hm * Hm:].float()).all()) and bool(torch.isnan(out[B * Ho * Ho:].float()).all())
    # the unfused kernels are the second opinion
    oa2 = torch.empty(B * Hm * Hm, 48, dtype=BF, device="cuda")
    ob2 = torch.empty(B * Hm * Hm, 48, dtype=BF, device="cuda")
    hip.stem_conv_dual(img, w, b, oa2, ob2)
    close(ob[:B * Hm * Hm], ob2, 1e-2, 1e-2)
    out2 = torch.empty(B * Ho * Ho, Cout, dtype=BF, device="cuda")
    hip.gemm(oa2, c2.weight, out2, M=B * Ho * Ho, N=Cout, bias=c2.bias, act=hip.ACT_RELU, conv=c2.geometry(), ktab=c2.ktab)
    close(out[:B * Ho * Ho], out2, 4e-2, 1e-2)


@pytest.mark.parametrize("C,k,g_", [(48, 16, 7), (96, 8, 7), (192, 4, 7), (768, 1, 7), (96, 4, 14)])
def test_dwpool(gpu_device, C, k, g_):
    B, H = 2, k * g_
    top = rnd(B, H, H, C, seed=37, dtype=BF)
    w = rnd(k * k, C, seed=38, scale=1.0 / k)
    out = torch.empty(B * g_ * g_, C, dtype=BF, device="cuda")
    hip.dwpool(top, w, out, B, H, H, C, k)
    ref = F.conv2d(top.float().permute(0, 3, 1, 2), w.t().reshape(C, 1, k, k), stride=k, groups=C).permute(0, 2, 3, 1)
    close(out.reshape(B, g_, g_, C), ref, 2e-2, 1e-2)


def test_lse_and_loss(gpu_device):
    R, N, off = 24, 72, 48
    rows = rnd(R, N, seed=39, scale=6.0)
    cols = rnd(R, N, seed=40, scale=6.0)
    lse = torch.empty(2, R, dtype=torch.float32, device="cuda")
    hip.lse_rows(rows, lse[0])
    hip.lse_rows(cols, lse[1])
    close(lse[0], torch.logsumexp(rows, 1), 1e-4)
    close(lse[1], torch.logsumexp(cols, 1), 1e-4)
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    hip.clip_loss_partial(lse[0], lse[1], rows, off, 1.0 / (2 * N), out)
    d = rows[torch.arange(R), off + torch.arange(R)]
    ref = ((torch.logsumexp(rows, 1) - d) + (torch.logsumexp(cols, 1) - d)).sum() / (2 * N)
    close(out[0], ref, 1e-4)
    odd = rnd(3, 5, seed=44, scale=4.0)
    l3 = torch.empty(3, dtype=torch.float32, device="cuda")
    hip.lse_rows(odd, l3)
    close(l3, torch.logsumexp(odd, 1), 1e-4)


@pytest.mark.parametrize("R,N,off,nsplit", [(24, 72, 48, 3), (512, 512, 0, 16), (100, 333, 200, 5), (32, 8192, 4096, 64),
                                            (512, 4096, 1536, 64), (1024, 8192, 7168, 64)])  # rank 3 of 8 at batch 512; rank 7 of 8 at 1024
def test_fused_lse_and_loss(gpu_device, R, N, off, nsplit):
    """MFMA GEMM + online log-sum-exp (no logits written) vs logsumexp of the explicit fp32 logits."""
    E, scale = 512, 14.285
    a = F.normalize(rnd(R, E, seed=50), dim=-1).to(BF)
    b = F.normalize(rnd(N, E, seed=51), dim=-1).to(BF)
    a2 = F.normalize(rnd(R, E, seed=52), dim=-1).to(BF)
    part = torch.full((4, R, nsplit), float("nan"), dtype=torch.float32, device="cuda")
    diag = torch.full((R,), float("nan"), dtype=torch.float32, device="cuda")
    hip.clip_lse_fused(a, b, scale, off, nsplit, part[0], part[1], diag)
    lg = scale * a.float() @ b.float().t()
    lse = torch.logsumexp(lg, 1)
    got = part[0].max(1).values + torch.log((part[1] * torch.exp(part[0] - part[0].max(1, keepdim=True).values)).sum(1))
    close(got, lse, 2e-3)
    close(diag, lg[torch.arange(R), off + torch.arange(R)], 2e-3)
    diag2 = torch.empty_like(diag)
    hip.clip_lse_fused(a2, b, scale, off, nsplit, part[2], part[3], diag2)
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    lse_out = torch.empty(2, R, dtype=torch.float32, device="cuda")
    hip.clip_loss_from_partials(part[0], part[1], part[2], part[3], diag, 1.0 / (2 * N), out, lse_out)
    lg2 = scale * a2.float() @ b.float().t()
    ref = ((lse - diag) + (torch.logsumexp(lg2, 1) - diag)).sum() / (2 * N)
    close(out[0], ref, 2e-3)
    close(lse_out[1], torch.logsumexp(lg2, 1), 2e-3)


def test_bad_arguments_are_rejected(gpu_device):
    x, w = rnd(8, 60, dtype=BF), rnd(8, 60, dtype=BF)             # K not a multiple of 64
    with pytest.raises(hip.HipError):
        hip.gemm(x, w, torch.empty(8, 8, dtype=BF, device="cuda"))
    with pytest.raises(hip.HipError):
        hip.attention(rnd(300, 2304, dtype=BF), torch.empty(300, 768, dtype=BF, device="cuda"), 1, 300, 12, False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_rows(gpu_device, dtype):
    """msclip_gather_rows: rows by stride or by index, fp32 and bf16, out of a wider parent (row-range / column views)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1000, 768, generator=g).to(dtype).cuda()
    out = torch.zeros(20, 768, dtype=dtype, device="cuda")
    hip.gather_rows(x, out, 20, row_mul=50)
    assert torch.equal(out, x[::50])
    idx = torch.randint(0, 1000, (33,), generator=g).to(torch.int32).cuda()
    out2 = torch.zeros(40, 768, dtype=dtype, device="cuda")
    hip.gather_rows(x, out2[3:36], 33, row_idx=idx)
    assert torch.equal(out2[3:36], x[idx.long()]) and not out2[:3].any() and not out2[36:].any()
    hip.gather_rows(x, out[:10], 10, row_mul=7, row_add=2)
    assert torch.equal(out[:10], x[2::7][:10])


def _dq(q, s):
    return q.view(hip.F8).float() * s[:, None]


@pytest.mark.parametrize("M,N,K", [(300, 768, 768), (1000, 2304, 1024), (2051, 1096, 768), (256 * 40 - 7, 3072, 1024), (512, 1024, 4096)])
def test_gemm_f8(gpu_device, M, N, K):
    """msclip_gemm_f8 (the ping-pong kernel on e4m3 operands, v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, per-row
    scales in the epilogue) against fp32 matmul of the DEQUANTISED operands: products of two e4m3 values are exact in fp32, so
    only the accumulation order differs.  Ragged M / N, bias, QuickGELU, fp32 residual, repeated launches bitwise equal."""
    xq, sx = hip.quantize_rows_f8(rnd(M, K, seed=81))
    wq, sw = hip.quantize_rows_f8(rnd(N, K, seed=82, scale=0.05))
    b = rnd(N, seed=83)
    ref = _dq(xq, sx) @ _dq(wq, sw).t() + b
    out = torch.full((M + 2, N), float("nan"), dtype=BF, device="cuda")
    hip.gemm_f8(xq, wq, out[:M], sx, sw, bias=b)
    close(out[:M], ref, 2e-2, 1e-2)
    assert bool(torch.isnan(out[M:].float()).all())
    o32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    hip.gemm_f8(xq, wq, o32, sx, sw, bias=b)
    close(o32, ref, 1e-3 * float(ref.abs().max()), 1e-4)
    hip.gemm_f8(xq, wq, out[:M], sx, sw, bias=b, act=hip.ACT_QUICKGELU)
    close(out[:M], ref * torch.sigmoid(1.702 * ref), 2e-2, 1e-2)
    r32 = rnd(M, N, seed=84)
    first = None
    for it in range(3):
        acc = r32.clone()
        hip.gemm_f8(xq, wq, acc, sx, sw, bias=b, resid=acc, resid_kind=hip.RESID_F32)
        if first is None:
            close(acc, r32 + ref, 1e-3 * float(ref.abs().max()), 1e-4)
            first = acc
        else:
            assert torch.equal(acc, first)


@pytest.mark.parametrize("C", [768, 1024])
def test_layernorm_f8_and_row_quant(gpu_device, C):
    """msclip_layernorm_f8: LayerNorm -> e4m3 + per-token scale (two parameter sets split by row) against torch; the scale is
    max |y| / 448 and the dequantised row is within e4m3's half-ulp (2^-4 relative; 2^-10 * 448 * s absolute near zero)."""
    M, split = 777, 300
    x = rnd(M, C, seed=91, scale=2.0)
    g1, b1, g2, b2 = rnd(C, seed=92) * 0.1 + 1, rnd(C, seed=93) * 0.1, rnd(C, seed=94) * 0.1 + 1, rnd(C, seed=95) * 0.1
    y = torch.cat([F.layer_norm(x[:split], (C,), g1, b1, 1e-12), F.layer_norm(x[split:], (C,), g2, b2, 1e-12)])
    q = torch.zeros(M, C, dtype=torch.uint8, device="cuda")
    s = torch.zeros(M, device="cuda")
    hip.layernorm_f8(x, g1, b1, g2, b2, split, q, s, M)
    close(s, y.abs().amax(1) / 448.0, 0.0, 1e-4)
    dq = _dq(q, s)
    tol = 0.0626 * y.abs() + (2.0 ** -9) * 448 * s[:, None]
    assert bool(((dq - y).abs() <= tol).all()), float(((dq - y).abs() - tol).max())
    xb = rnd(M, C, seed=96, dtype=BF)
    hip.quant_f8_rows(xb, q, s)
    close(s, xb.float().abs().amax(1) / 448.0, 0.0, 1e-5)
    ref_q, _ = hip.quantize_rows_f8(xb)                                          # torch's own e4m3 rounding of the same scaled values
    assert (q != ref_q).float().mean().item() < 1e-3                              # (1/s vs division: a rare last-bit tie)


@pytest.mark.parametrize("M,N,K,resid", [(256 * 100, 2304, 768, False), (256 * 100 + 37, 768, 768, True), (256 * 90, 3072, 256, False),
                                          (256 * 33 + 5, 3072 - 8, 1024, False)])
def test_gemm_chip_filling_launches_back_to_back_and_on_two_streams(gpu_device, M, N, K, resid):
    """Launches of >= 3 tiles per workgroup, repeated and on two streams at once: every tile exactly once (NaN prefill: a
    skipped tile stays NaN; a tile computed twice into the fp32 residual stream doubles its increment).  Written for the
    dynamic per-XCD tile lists that were tried in round 3 (history: 'gemm: dynamic tile lists'), kept for the static ones."""
    x, w, b = rnd(M, K, seed=41, dtype=BF), rnd(N, K, seed=42, scale=0.05, dtype=BF), rnd(N, seed=43)
    ref = x.float() @ w.float().t() + b
    outs = []
    for rep in range(6):
        if resid:
            out = torch.ones(M, N, dtype=torch.float32, device="cuda")
            hip.gemm(x, w, out, bias=b, resid=out, resid_kind=hip.RESID_F32)
        else:
            out = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
            hip.gemm(x, w, out, bias=b)
        outs.append(out)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for rep in range(3):
            o2 = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
            hip.gemm(x, w, o2, bias=b)
            outs.append(o2)
    for rep in range(3):
        o3 = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
        hip.gemm(x, w, o3, bias=b)
        outs.append(o3)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for out in outs:
        if out.dtype == torch.float32:
            close(out, ref + 1.0, 6e-2, 1e-2)
        else:
            close(out, ref, 6e-2, 2e-2)


@pytest.mark.parametrize("tile", [0, 4])
@pytest.mark.parametrize("M,N,K", [(768, 520, 128), (2048, 1096, 768), (256 * 9, 3072, 768)])
def test_gemm_training_epilogue_forms(gpu_device, tile, M, N, K):
    """The ping-pong kernels' training-step epilogues: out2 = the value BEFORE the activation beside QuickGELU(value) from one
    launch (c_fc's forward), and resid_kind 4 = multiply by QuickGELU'(h) (c_proj's dgrad); whole 256-row tiles (any other
    M is rejected: the guarded edge epilogue does not carry these forms), ragged N."""
    x, w, b = rnd(M, K, seed=31, dtype=BF), rnd(N, K, seed=32, scale=0.05, dtype=BF), rnd(N, seed=33)
    pre = x.float() @ w.float().t() + b
    out = torch.full((M + 1, N), float("nan"), dtype=BF, device="cuda")
    out2 = torch.full((M + 1, N), float("nan"), dtype=BF, device="cuda")
    hip.gemm(x, w, out[:M], bias=b, act=hip.ACT_QUICKGELU, out2=out2[:M], tile=tile)
    close(out2[:M], pre, 2e-2, 1e-2)
    close(out[:M], pre * torch.sigmoid(1.702 * pre), 2e-2, 1e-2)
    assert bool(torch.isnan(out[M:].float()).all()) and bool(torch.isnan(out2[M:].float()).all())
    h = rnd(M, N, seed=34, scale=1.5, dtype=BF)
    s = torch.sigmoid(1.702 * h.float())
    ref = (x.float() @ w.float().t()) * (s + 1.702 * h.float() * s * (1 - s))
    dh = torch.full((M + 1, N), float("nan"), dtype=BF, device="cuda")
    hip.gemm(x, w, dh[:M], resid=h, resid_kind=hip.RESID_GELUGRAD, tile=tile)
    close(dh[:M], ref, 2e-2, 1e-2)
    assert bool(torch.isnan(dh[M:].float()).all())
    # ... and with colsum_part the same launch leaves the stored values' column sums per 128 rows (c_fc's bias gradient without a
    # second pass over dh): same dh bit for bit; partials = the fp32 values' sums, i.e. the bf16 output's up to its rounding
    dh2 = torch.full((M + 1, N), float("nan"), dtype=BF, device="cuda")
    part = torch.full((M // 128 + 1, N), float("nan"), dtype=torch.float32, device="cuda")
    hip.gemm(x, w, dh2[:M], resid=h, resid_kind=hip.RESID_GELUGRAD, tile=tile, colsum_part=part[:M // 128])
    assert torch.equal(dh2[:M], dh[:M]) and bool(torch.isnan(dh2[M:].float()).all())
    assert bool(torch.isnan(part[M // 128:]).all())
    want = dh[:M].float().view(M // 128, 128, N).sum(1)
    scale = dh[:M].float().abs().view(M // 128, 128, N).sum(1)
    assert float(((part[:M // 128] - want).abs() / (scale * 2.0 ** -8 + 1e-6)).max()) < 1.0      # 128 roundings of <= 2^-8 relative each (bf16 unit roundoff)
    close(part[:M // 128], ref.view(M // 128, 128, N).sum(1), 2e-2, 1e-1)
    close(hip.colsum(part[:M // 128]), ref.sum(0), 2e-2, 3e-1)
    assert hip.gemm_variant(hip.describe_gemm(0, M, N, K, tile=1, resid_kind=hip.RESID_GELUGRAD)) == "invalid"   # other kernels reject it
    assert hip.gemm_variant(hip.describe_gemm(0, M + 8, N, K, resid_kind=hip.RESID_GELUGRAD)) == "invalid"       # ... and so do ragged M


@pytest.mark.parametrize("M,N,K", [(512, 1024, 768), (256 * 5, 4096, 1024)])
def test_gemm_f8_e4m3_output(gpu_device, M, N, K):
    """msclip_gemm_f8 with an e4m3 OUTPUT (c_fc -> the fp8 operand of c_proj): out = fp8(QuickGELU(acc + bias) * out_scale),
    saturating.  Against torch's own e4m3 rounding of the fp32 reference: equal up to one fp8 ulp where the fp32 accumulation
    order moves a value across a rounding boundary, saturation at +-448."""
    xq, sx = hip.quantize_rows_f8(rnd(M, K, seed=85))
    wq, sw = hip.quantize_rows_f8(rnd(N, K, seed=86, scale=0.05))
    b = rnd(N, seed=87)
    pre = _dq(xq, sx) @ _dq(wq, sw).t() + b
    ref = pre * torch.sigmoid(1.702 * pre)
    s = float(ref.abs().max()) / 448.0 * 2.0                       # half of the range: the top values saturate
    out = torch.zeros(M, N, dtype=torch.uint8, device="cuda")
    hip.gemm_f8(xq, wq, out, sx, sw, bias=b, act=hip.ACT_QUICKGELU, out_scale=1.0 / (s / 2.0))
    got = out.view(hip.F8).float() * (s / 2.0)
    want = (ref / (s / 2.0)).clamp(-448, 448).to(hip.F8).float() * (s / 2.0)
    err = (got - want).abs()
    tol = 0.126 * want.abs() + (2.0 ** -9) * 448 * (s / 2.0) + 1e-3 * float(ref.abs().max())     # one e4m3 ulp is 1/16 .. 1/8 of the value
    print("e4m3 output: elements differing from torch's rounding of the fp32 reference:", (got != want).float().mean().item(),
          "worst err / tol", float((err / tol).max()))
    assert bool((err <= tol).all()), float((err - tol).max())
    assert (got != want).float().mean().item() < 0.02               # nearly all elements bit-equal
    assert float(got.abs().max()) <= 448 * (s / 2.0) + 1e-6 and bool((got.abs() >= 447 * (s / 2.0)).any())    # saturated, no NaN


# ---------------------------------------------------------------------------------------------------------------------
# round 4: LayerNorm fold (DESIGN.md "LayerNorm fold"): out_proj / c_proj produce the next LayerNorm's operands, the
# projection behind it applies the normalisation to its accumulators
# ---------------------------------------------------------------------------------------------------------------------

def _fold_weights(w, bias, gamma, beta):
    """W' = bf16(gamma o W), csum = row sums of W' (of the bf16 values), bias' = bias + W beta (fp32)."""
    wf = (w.float() * gamma[None, :]).to(BF).contiguous()
    return wf, wf.float().sum(dim=1).contiguous(), (bias + w.float() @ beta).contiguous()


@pytest.mark.parametrize("M,K", [(1024, 768), (2560, 3072)])
def test_layernorm_fold_producer(gpu_device, M, K):
    """msclip_gemm with xb / center / part: the fp32 residual update is bitwise the plain launch's, xb is bitwise
    bf16(out - center), the partial sums are the 64-column sums of (out - center) and its square; msclip_rowstat_finalize
    turns them into (rstd, mean * rstd) of M.py:204-219 and moves the centre to the row mean."""
    D = 768
    a, w, b = rnd(M, K, seed=71, dtype=BF), rnd(D, K, seed=72, scale=0.03, dtype=BF), rnd(D, seed=73)
    x0 = rnd(M, D, seed=74) + rnd(M, 1, seed=75, scale=3.0)               # rows with a sizeable mean
    cen = x0.mean(dim=1) + rnd(M, seed=76, scale=0.2)                      # "mean at the previous LayerNorm point"
    plain = x0.clone()
    hip.gemm(a, w, plain, bias=b, resid=plain, resid_kind=hip.RESID_F32, tile=4)          # the same main loop
    out, xb = x0.clone(), torch.full((M, D), float("nan"), dtype=BF, device="cuda")
    part = torch.full((M, D // 64, 2), float("nan"), device="cuda")
    c = cen.clone()
    fo = hip.FoldOut(xb, c, part)
    for rep in range(2):
        out.copy_(x0)
        hip.gemm(a, w, out, bias=b, resid=out, resid_kind=hip.RESID_F32, fold_out=fo)
        assert torch.equal(out, plain)
        assert torch.equal(xb.view(torch.int16), (out - cen[:, None]).to(BF).view(torch.int16))
    d = (out - cen[:, None]).view(M, D // 64, 64)
    close(part[..., 0], d.sum(-1), 2e-3, 1e-5)
    close(part[..., 1], (d * d).sum(-1), 2e-3, 1e-5)
    rstat = torch.empty(M, 2, device="cuda")
    hip.rowstat_finalize(part, c, rstat, M, D)
    mu, var = out.mean(dim=1), out.var(dim=1, unbiased=False)
    close(c, mu, 1e-5, 1e-6)
    close(rstat[:, 0], 1.0 / torch.sqrt(var + 1e-12), 0.0, 2e-5)
    close(rstat[:, 1], (mu - cen) / torch.sqrt(var + 1e-12), 2e-5, 2e-5)


@pytest.mark.parametrize("act", [hip.ACT_NONE, hip.ACT_QUICKGELU])
@pytest.mark.parametrize("two", [False, True])
def test_layernorm_fold_consumer(gpu_device, act, two):
    """The projection behind a folded LayerNorm against fp32 torch (LayerNorm of M.py:204-219, then the linear layer) and
    against the unfused chain of this library (msclip_layernorm_split + msclip_gemm): both deviate from fp32 only by the bf16
    rounding of their operands, and the fold's deviation must not exceed the unfused chain's by more than a quarter.  Two row
    segments = image / text rows with their own gamma / beta under one shared weight."""
    M, D, N = 2048, 768, 2304 if act == hip.ACT_NONE else 3072
    split = 768 if two else M
    x = rnd(M, D, seed=81, scale=1.5) + rnd(M, 1, seed=82, scale=2.0)
    w, b = rnd(N, D, seed=83, scale=0.04, dtype=BF), rnd(N, seed=84, scale=0.3)
    g1, b1 = 1.0 + rnd(D, seed=85, scale=0.2), rnd(D, seed=86, scale=0.3)
    g2, b2 = 1.0 + rnd(D, seed=87, scale=0.2), rnd(D, seed=88, scale=0.3)
    mu, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    xn = (x - mu) / torch.sqrt(var + 1e-12)
    ln = torch.cat([xn[:split] * g1 + b1, xn[split:] * g2 + b2])
    ref = ln @ w.float().t() + b
    if act == hip.ACT_QUICKGELU:
        ref = ref * torch.sigmoid(1.702 * ref)
    # unfused chain
    lno = torch.empty(M, D, dtype=BF, device="cuda")
    hip.layernorm_split(x, g1, b1, g2, b2, split, lno, M)
    chain = torch.empty(M, N, dtype=BF, device="cuda")
    hip.gemm(lno, w, chain, bias=b, act=act, tile=4)
    # fold: centres a little off the true means, as they are one layer later
    cen = mu[:, 0] + rnd(M, seed=89, scale=0.05)
    xb = (x - cen[:, None]).to(BF).contiguous()
    d = x - cen[:, None]
    mu_d = d.mean(1)
    rstd = 1.0 / torch.sqrt(d.var(1, unbiased=False) + 1e-12)
    rstat = torch.stack([rstd, mu_d * rstd], dim=1).contiguous()
    w1, c1, bb1 = _fold_weights(w, b, g1, b1)
    w2, c2, bb2 = _fold_weights(w, b, g2, b2)
    fi = hip.FoldIn(rstat, c1, w2, bb2, c2, split) if two else hip.FoldIn(rstat, c1)
    out = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
    outs = []
    for rep in range(2):
        hip.gemm(xb, w1, out, bias=bb1, act=act, fold_in=fi)
        outs.append(out.clone())
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    e_fold = (out.float() - ref).abs()
    e_chain = (chain.float() - ref).abs()
    scale = ref.abs().max().item()
    assert e_fold.max().item() <= 0.02 * scale, (e_fold.max().item(), scale)
    assert e_fold.mean().item() <= 1.25 * e_chain.mean().item() + 1e-4, (e_fold.mean().item(), e_chain.mean().item())
    # a descriptor the fold cannot take (ragged rows) is rejected, not mis-computed
    with pytest.raises(hip.HipError):
        hip.gemm(xb[:1000], w1, out[:1000], bias=bb1, act=act, fold_in=hip.FoldIn(rstat, c1))


def test_layernorm_stats_rows(gpu_device):
    """msclip_layernorm_stats = msclip_layernorm + the fold's per-row state (centre = mean, rowstat = (1, 0)), with the raw copy."""
    M, D = 1000, 768
    x, g, b = rnd(M, D, seed=91) + 2.0, 1.0 + rnd(D, seed=92, scale=0.1), rnd(D, seed=93, scale=0.1)
    ref = torch.empty(M, D, dtype=BF, device="cuda")
    hip.layernorm(x, g, b, ref, M)
    out, raw = torch.empty(M, D, dtype=BF, device="cuda"), torch.empty(M, D, device="cuda")
    cen, rs = torch.empty(M, device="cuda"), torch.empty(M, 2, device="cuda")
    hip.layernorm_stats(x, g, b, out, M, cen, rs, raw_out=raw)
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16)) and torch.equal(raw, x)
    close(cen, x.mean(1), 1e-5)
    assert bool((rs[:, 0] == 1).all()) and bool((rs[:, 1] == 0).all())


@pytest.mark.parametrize("M,N,K,out_f32", [(8192, 96, 128, False), (5000, 48, 64, True), (4100, 192, 192, False)])
def test_stream_gemm_row_scatter(gpu_device, M, N, K, out_f32):
    """The streaming small-K kernel with the descriptor's row scatter (store row m + (m / rpg) * radd + roff; round 4: the input
    gradients of the stride-2 convolutions scatter their parity classes this way).  Rows that are not store targets stay untouched."""
    x, w = rnd(M, K, seed=3, dtype=BF), rnd(N, K, seed=4, scale=0.1, dtype=BF)
    assert hip.gemm_variant(hip.describe_gemm(0, M, N, K, 0, None, rpg=49)) == "stream"
    rows = M + (M // 49) * 3 + 2
    out = torch.full((rows, N), float("nan"), dtype=torch.float32 if out_f32 else BF, device="cuda")
    hip.gemm(x, w, out, M=M, rpg=49, radd=3, roff=2)
    m = torch.arange(M, device="cuda")
    tgt = m + (m // 49) * 3 + 2
    ref = x.float() @ w.float().t()
    close(out[tgt], ref, 2e-2 if not out_f32 else 1e-3, 1e-2 if not out_f32 else 1e-4)
    rest = torch.ones(rows, dtype=torch.bool, device="cuda")
    rest[tgt] = False
    assert bool(torch.isnan(out[rest].float()).all())
    if not out_f32:
        # resid_kind 6: the launch adds what is already at the store row (in place): the stride-2 shortcuts' input gradients
        base = rnd(rows, N, seed=5, dtype=BF)
        acc = base.clone()
        hip.gemm(x, w, acc, M=M, rpg=49, radd=3, roff=2, resid=acc, resid_kind=hip.RESID_ACCUM)
        close(acc[tgt], ref + base[tgt].float(), 3e-2, 1e-2)
        assert torch.equal(acc[rest], base[rest])


def test_layernorm_fold_producer_second_residual_stream(gpu_device):
    """msclip_gemm_desc.resid2: the rows from seg_split on read their residual from a second matrix (the text rows' stream is
    `out`, the image rows' sits in the lateral adapter's output buffer): bitwise the one-stream launch on the joined rows."""
    M, Mv, D, K = 1536, 512, 768, 768
    a, w, b = rnd(M, K, seed=71, dtype=BF), rnd(D, K, seed=72, scale=0.03, dtype=BF), rnd(D, seed=73)
    x0, cen = rnd(M, D, seed=74) + 1.5, rnd(M, seed=76, scale=0.2) + 1.5
    ref, xb0, p0 = x0.clone(), torch.empty(M, D, dtype=BF, device="cuda"), torch.empty(M, D // 64, 2, device="cuda")
    hip.gemm(a, w, ref, bias=b, resid=ref, resid_kind=hip.RESID_F32, fold_out=hip.FoldOut(xb0, cen.clone(), p0))
    xa = x0[:Mv].clone()                                                    # image rows' stream
    out = x0.clone()
    out[:Mv] = float("nan")                                                # ... and nothing of it in `out`
    xb, part = torch.empty_like(xb0), torch.empty_like(p0)
    hip.gemm(a, w, out, bias=b, resid=xa, resid_kind=hip.RESID_F32, fold_out=hip.FoldOut(xb, cen.clone(), part, resid2=out, split=Mv))
    assert torch.equal(out, ref) and torch.equal(xb.view(torch.int16), xb0.view(torch.int16)) and torch.equal(part, p0)
    assert torch.equal(xa, x0[:Mv])
    with pytest.raises(AssertionError):                                    # a split inside a tile is rejected
        hip.gemm(a, w, out, bias=b, resid=xa, resid_kind=hip.RESID_F32, fold_out=hip.FoldOut(xb, cen.clone(), part, resid2=out, split=300))


@pytest.mark.parametrize("form", ["sample", "gridrow"])
@pytest.mark.parametrize("B,g_,C,usecls", [(5, 7, 768, True), (3, 14, 768, False), (2, 16, 1024, True)])
def test_adapter_combine_ln_stats(gpu_device, monkeypatch, B, g_, C, usecls, form):
    """msclip_adapter_combine_ln_stats = msclip_adapter_combine_ln followed by msclip_layernorm_stats of its output (the second
    LayerNorm runs on the fp32 values the first one stores); both launch forms (workgroup per sample / wave per grid row)."""
    monkeypatch.setenv("MSCLIP_ADAPTER_SAMPLE", "1" if form == "sample" else "0")
    L = g_ * g_ + 1
    x, t = rnd(B * L, C, seed=61), rnd(B * g_ * g_, C, seed=62)
    dww, dwb = rnd(9, C, seed=63, scale=0.3), rnd(C, seed=64, scale=0.1)
    ga, be = 1.0 + rnd(C, seed=65, scale=0.1), rnd(C, seed=66, scale=0.1)
    g1, b1 = 1.0 + rnd(C, seed=67, scale=0.1), rnd(C, seed=68, scale=0.1)
    xa0 = torch.empty(B * L, C, device="cuda")
    hip.adapter_combine_ln(x, t, dww, dwb, ga, be, xa0, B, L, g_, usecls)
    l0, c0, r0 = torch.empty(B * L, C, dtype=BF, device="cuda"), torch.empty(B * L, device="cuda"), torch.empty(B * L, 2, device="cuda")
    hip.layernorm_stats(xa0, g1, b1, l0, B * L, c0, r0)
    xa = torch.full((B * L + 1, C), float("nan"), device="cuda")
    lno = torch.full((B * L + 1, C), float("nan"), dtype=BF, device="cuda")
    cen, rs = torch.full((B * L + 1,), float("nan"), device="cuda"), torch.full((B * L + 1, 2), float("nan"), device="cuda")
    hip.adapter_combine_ln_stats(x, t, dww, dwb, ga, be, xa[:B * L], g1, b1, lno[:B * L], cen, rs, B, L, g_, usecls)
    # the fp32 stream is bitwise; the second LayerNorm is the same code inlined into another kernel, and under -ffast-math the
    # compiler orders its sums per kernel: last-bit differences of the statistics, i.e. rare one-ulp differences of the bf16 output
    assert torch.equal(xa[:B * L], xa0) and torch.equal(rs[:B * L], r0)
    close(cen[:B * L], c0, 1e-6, 1e-6)
    a, b = lno[:B * L].float(), l0.float()
    assert float(((a - b).abs() / b.abs().clamp_min(2.0 ** -6)).max()) <= 2.0 ** -7 and float((a != b).float().mean()) < 5e-3
    assert bool(torch.isnan(xa[B * L:]).all()) and bool(torch.isnan(lno[B * L:].float()).all()) and bool(torch.isnan(cen[B * L:]).all())


def test_adapter_sample_form_against_the_gridrow_form(gpu_device, monkeypatch):
    """The workgroup-per-sample adapter kernel does the grid-row kernel's arithmetic in the same order: the fp32 stream bitwise."""
    B, g_, C = 37, 7, 768
    L = g_ * g_ + 1
    x, t = rnd(B * L, C, seed=71), rnd(B * g_ * g_, C, seed=72)
    dww, dwb = rnd(9, C, seed=73, scale=0.3), rnd(C, seed=74, scale=0.1)
    ga, be = 1.0 + rnd(C, seed=75, scale=0.1), rnd(C, seed=76, scale=0.1)
    g1, b1 = 1.0 + rnd(C, seed=77, scale=0.1), rnd(C, seed=78, scale=0.1)
    res = {}
    for form in ("1", "0"):
        monkeypatch.setenv("MSCLIP_ADAPTER_SAMPLE", form)
        xa, lno = torch.empty(B * L, C, device="cuda"), torch.empty(B * L, C, dtype=BF, device="cuda")
        cen, rs = torch.empty(B * L, device="cuda"), torch.empty(B * L, 2, device="cuda")
        hip.adapter_combine_ln_stats(x, t, dww, dwb, ga, be, xa, g1, b1, lno, cen, rs, B, L, g_, True)
        plain = torch.empty(B * L, C, device="cuda")
        hip.adapter_combine_ln(x, t, dww, dwb, ga, be, plain, B, L, g_, True)
        res[form] = (xa, lno.float(), cen, rs, plain)
    (xa1, l1, c1, r1, p1), (xa0, l0, c0, r0, p0) = res["1"], res["0"]
    # within a form the fp32 stream of the stats launch is bitwise the plain launch's; across forms the same sums are ordered per
    # kernel under -ffast-math: last-bit differences of the fp32 stream and the statistics, rare one-ulp differences of the bf16 operand
    assert torch.equal(xa1, p1) and torch.equal(xa0, p0) and torch.equal(r1, r0)
    close(xa1, xa0, 5e-6, 5e-6)
    close(c1, c0, 5e-6, 5e-6)
    assert float(((l1 - l0).abs() / l0.abs().clamp_min(2.0 ** -6)).max()) <= 2.0 ** -7 and float((l1 != l0).float().mean()) < 5e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,S,P", [(3, 224, 14), (2, 224, 16), (1, 64, 32)])
def test_patchify_and_patch_conv(gpu_device, dtype, B, S, P):
    """msclip_patchify + msclip_gemm with the token scatter = the plain patch convolution + cls / positional layout of
    M.py:2657-2664 (BASELINE config C5's stem): the patch matrix is bitwise the bf16 unfold of the image (pad columns zero),
    the GEMM over it against F.conv2d."""
    g = S // P
    kp, D = 3 * P * P, 256
    kpad = (kp + 63) // 64 * 64
    img = rnd(B, 3, S, S, seed=95).to(dtype)
    pm = torch.full((B * g * g + 2, kpad), float("nan"), dtype=BF, device="cuda")
    hip.patchify(img, pm[:B * g * g], B, S, P, kpad)
    ref = F.unfold(img.float(), P, stride=P).transpose(1, 2).reshape(B * g * g, kp).to(BF)      # columns (c, kh, kw), rows (b, py, px)
    assert torch.equal(pm[:B * g * g, :kp].view(torch.int16), ref.view(torch.int16))
    assert bool((pm[:B * g * g, kp:] == 0).all()) and bool(torch.isnan(pm[B * g * g:].float()).all())
    w = rnd(D, 3, P, P, seed=96, scale=0.05)
    wp = torch.zeros(D, kpad, dtype=BF, device="cuda")
    wp[:, :kp] = w.reshape(D, kp).to(BF)
    pos = rnd(g * g + 1, D, seed=97)
    L = g * g + 1
    X = torch.zeros(B * L, D, device="cuda")
    hip.gemm(pm[:B * g * g], wp, X, M=B * g * g, resid=pos, resid_kind=hip.RESID_TABLE, rpg=g * g, radd=1, roff=1)
    conv = F.conv2d(img.float().to(BF).float(), w.to(BF).float(), stride=P).flatten(2).transpose(1, 2)           # [B, g*g, D]
    close(X.view(B, L, D)[:, 1:], conv + pos[1:], 2e-2, 1e-3)
    assert bool((X.view(B, L, D)[:, 0] == 0).all())                                              # the cls rows are not the GEMM's

@pytest.mark.parametrize("case", ["image", "captions", "both_fold"])
def test_fused_qkv_attention_against_gemm_plus_attention(gpu_device, case):
    """msclip_qkv_attention (in_proj + attention in one kernel, q|k|v staged in LDS, block-diagonal / causal mask from the row
    table) against the two-launch chain it replaces on the same inputs: msclip_gemm (+ the LayerNorm fold's consumer form) into a
    q|k|v matrix, then msclip_attention / msclip_attention_varlen.  Image samples of 50 tokens (tiles of 5 samples, a partial last
    tile), packed captions of 3..77 rows (causal), and both modalities in one launch with per-modality folded weights."""
    H, D = 12, 768
    Bi, Lv = (0, 50) if case == "captions" else (128, 50) if case == "both_fold" else (23, 50)   # 128 x 50 = 25 whole GEMM tiles
    Bt = 0 if case == "image" else 41
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(3, 78, (Bt,), generator=g).tolist() if Bt else []
    if Bt:
        lens[0], lens[1] = 77, 3
    starts = [i * Lv for i in range(Bi)]
    Mv = Bi * Lv
    r = Mv
    for n in lens:
        starts.append(r)
        r += n
    live = r
    M = -(-live // 256) * 256 if case == "both_fold" else live
    cu = torch.tensor(starts + [live], dtype=torch.int32, device="cuda")
    x = rnd(M, D, seed=81, dtype=BF)
    w, b = rnd(3 * D, D, seed=82, scale=0.03, dtype=BF), rnd(3 * D, seed=83, scale=0.1)
    fold = case == "both_fold"
    qkv = torch.empty(M, 3 * D, dtype=BF, device="cuda")
    if fold:
        w2, b2 = rnd(3 * D, D, seed=84, scale=0.03, dtype=BF), rnd(3 * D, seed=85, scale=0.1)
        rstat = torch.stack([1.0 + 0.1 * torch.rand(M, generator=g), 0.1 * torch.randn(M, generator=g)], 1).cuda().contiguous()
        c1, c2 = w.float().sum(1).contiguous(), w2.float().sum(1).contiguous()
        split = Mv                                            # the modality boundary: a whole GEMM tile for the two-launch form
        assert split % 256 == 0
        hip.gemm(x, w, qkv, bias=b, fold_in=hip.FoldIn(rstat, c1, w2, b2, c2, split))
    else:
        hip.gemm(x, w, qkv, bias=b)
    ref = torch.zeros(M, D, dtype=BF, device="cuda")
    if Bi:
        hip.attention(qkv[:Mv], ref[:Mv], Bi, Lv, H, False)
    if Bt:
        cut = torch.tensor([s - Mv for s in starts[Bi:]] + [live - Mv, max(lens)], dtype=torch.int32, device="cuda")
        hip.attention_varlen(qkv[Mv:], ref[Mv:], cut, Bt, max(lens), H, True)
    tabs = hip.QkvAttnTables(cu, Bi + Bt, split_sample=Bi if (Bi and Bt) else 0, total_rows=live)
    torch.cuda.synchronize()
    nt = int(tabs.ntiles)
    assert nt > 0
    tf = tabs.tile_first[:nt + 1].tolist()
    cul = cu.tolist()
    assert tf[0] == 0 and tf[-1] == Bi + Bt and all(cul[tf[i + 1]] - cul[tf[i]] <= 256 for i in range(nt))
    assert not (Bi and Bt) or Bi in tf                            # no tile straddles the modality boundary
    out = torch.full((M + 1, D), float("nan"), dtype=BF, device="cuda")
    wh, bh = hip.head_major_qkv(w, b, H)
    fi = None
    if fold:
        w2h, b2h, c2h = hip.head_major_qkv(w2, b2, H, c2)
        _, _, c1h = hip.head_major_qkv(w, b, H, c1)
        fi = hip.FoldIn(rstat, c1h, w2h, b2h, c2h, split)
    hip.qkv_attention(x, wh, bh, out[:M], tabs, H, causal_from_row=Mv if Bt else hip.INT_MAX, fold_in=fi, M=M)
    torch.cuda.synchronize()
    a, e = out[:live].float(), ref[:live].float()
    assert bool(torch.isfinite(a).all())
    # both paths round q|k|v to bf16 (the fused one from the same fp32 accumulators, summed in another order) and P to bf16
    err = (a - e).abs()
    assert float(err.max()) <= 0.03 * float(e.abs().max()) + 2e-3, (case, float(err.max()), float(e.abs().max()))
    assert float(err.mean()) <= 2e-3 * float(e.abs().mean()) + 2e-4
    assert bool(torch.isnan(out[M:].float()).all())
    again = torch.empty(M, D, dtype=BF, device="cuda")
    hip.qkv_attention(x, wh, bh, again, tabs, H, causal_from_row=Mv if Bt else hip.INT_MAX, fold_in=fi, M=M)
    assert torch.equal(again[:live].view(torch.int16), out[:live].view(torch.int16))      # race screen: bitwise repeatable
