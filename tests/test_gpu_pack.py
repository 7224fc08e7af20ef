"""Packed (pad-free) captions on a real MI355X.

Under the causal mask (reference M.py:2965-2971) a row behind a caption's EOT position cannot reach the EOT row that encode_text
returns (M.py:3057-3060) in any block, so the engine gives caption b only n_b = argmax + 1 rows.  Here: the C-ABI entry points of
the packed layout against plain fp32 torch on the per-caption slices, and the engine with packing on against (a) the
full-row path (MSCLIP_TEXT_PACK=0), (b) the oracle, on ragged batches with the edge captions the reference's semantics create:
EOT at position 1 (one content-free caption: 2 live rows), a 75-token caption (77 live rows), a second 49407 behind the first
(torch.argmax takes the first), an EOT id at position 0.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import set_opt
from msclip_amd import hip, synth

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


def edge_tokens(B, seed=3, L=77, vocab=49408):
    """synth captions + the edge cases in the first rows."""
    tok = synth.synth_tokens(B, seed=seed).clone()
    sot, eot = vocab - 2, vocab - 1
    if B >= 1:
        tok[0] = 0
        tok[0, 0], tok[0, 1] = sot, eot                                         # no content: EOT at position 1 (2 live rows)
    if B >= 2:
        tok[1, 0] = sot
        tok[1, 1:76] = torch.randint(1, sot, (75,), generator=torch.Generator().manual_seed(seed))
        tok[1, 76] = eot                                                        # 75 content tokens: all 77 rows live
    if B >= 3:
        n = int((tok[2] == eot).nonzero()[0])
        tok[2, min(n + 5, L - 1)] = eot                                         # a second EOT id behind the first: argmax takes the first
    if B >= 4:
        tok[3] = 0
        tok[3, 0] = eot                                                         # the maximum at position 0: ONE live row
    return tok


def lengths(tok):
    return (tok.argmax(dim=-1) + 1).to(torch.int32)


def cu_of(n):
    cu = torch.zeros(n.numel() + 2, dtype=torch.int32)
    cu[1:n.numel() + 1] = torch.cumsum(n, 0)
    cu[n.numel() + 1] = n.max()
    return cu


@pytest.mark.parametrize("B", [1, 5, 64, 1500])
def test_text_lengths_and_packed_embedding(gpu_device, B):
    L, C, V, base = 77, 768, 49408, 11
    tok = edge_tokens(B).cuda()
    n = lengths(tok.cpu())
    ln = torch.full((B + 3,), -7, dtype=torch.int32, device="cuda")
    cu = torch.full((B + 5,), -7, dtype=torch.int32, device="cuda")
    eot = torch.full((B + 3,), -7, dtype=torch.int32, device="cuda")
    hip.text_lengths(tok, ln, cu, eot, row_base=base)
    ref = cu_of(n)
    assert torch.equal(ln[:B].cpu(), n) and bool((ln[B:] == -7).all())
    assert torch.equal(cu[:B + 2].cpu(), ref) and bool((cu[B + 2:] == -7).all())
    assert torch.equal(eot[:B].cpu(), base + ref[:B] + n - 1) and bool((eot[B:] == -7).all())
    total = int(ref[B])
    padded = -(-total // 256) * 256 if -(-total // 256) * 256 <= B * L else total
    emb, pos = rnd(V, C, seed=1, scale=0.02), rnd(L, C, seed=2, scale=0.01)
    x = torch.full((base + B * L + 4, C), 7.0, device="cuda")
    hip.embed_tokens_packed(tok, emb, pos, x, cu, base, padded)
    for b in ([0, 1, 2, 3, B - 1] if B >= 5 else range(B)):
        r0, k = base + int(ref[b]), int(n[b])
        assert torch.equal(x[r0:r0 + k], emb[tok[b, :k]] + pos[:k]), b
    assert bool((x[:base] == 7.0).all()) and bool((x[base + total:base + padded] == 0).all()) and bool((x[base + padded:] == 7.0).all())
    # every live row, in one comparison
    bidx = torch.repeat_interleave(torch.arange(B), n.long())
    lidx = torch.cat([torch.arange(int(k)) for k in n])
    assert torch.equal(x[base:base + total], emb[tok[bidx, lidx]] + pos[lidx.cuda()])


def _packed_qkv(n, D, seed, pad, base=0):
    total = int(n.sum())
    return rnd(base + total + pad + 3, 3 * D, seed=seed, dtype=BF)


@pytest.mark.parametrize("lens,causal", [([2, 77, 31, 1, 33, 64, 65, 50], True), ([5, 9, 32, 1, 17], True), ([40, 64, 3], False),
                                         ([96, 2, 70], True)])
def test_attention_varlen(gpu_device, lens, causal):
    Hh, D = 12, 768
    n = torch.tensor(lens, dtype=torch.int32)
    B, total, pad = len(lens), int(n.sum()), 9
    cu = cu_of(n).cuda()
    qkv = _packed_qkv(n, D, 16, pad)
    out = torch.full((total + pad + 3, D), 7.0, dtype=BF, device="cuda")
    hip.attention_varlen(qkv, out, cu, B, int(n.max()), Hh, causal, pad_rows=pad)
    for b in range(B):
        r0, L = int(cu[b]), lens[b]
        q, k, v = qkv[r0:r0 + L].float().reshape(L, 3, Hh, 64).permute(1, 2, 0, 3)
        s = q @ k.transpose(-1, -2)
        if causal:
            s = s + torch.full((L, L), float("-inf"), device="cuda").triu_(1)
        ref = (torch.softmax(s, -1) @ v).permute(1, 0, 2).reshape(L, D)
        close(out[r0:r0 + L], ref, 2e-2, 2e-2)
        # the fixed-length kernel on this caption alone: same arithmetic
        one = torch.empty(L, D, dtype=BF, device="cuda")
        hip.attention(qkv[r0:r0 + L], one, 1, L, Hh, causal)
        close(out[r0:r0 + L], one, 2e-2, 1e-2)
    assert bool((out[total:total + pad] == 0).all()) and bool((out[total + pad:] == 7.0).all())
    assert hip.lib().msclip_attention_varlen(qkv.data_ptr(), out.data_ptr(), None, B, 77, Hh, 3 * D, D, 1, 0, None, None) == -1
    assert hip.lib().msclip_attention_varlen(qkv.data_ptr(), out.data_ptr(), cu.data_ptr(), B, 97, Hh, 3 * D, D, 1, 0, None, None) == -1


def test_attention_lastq_varlen(gpu_device):
    Hh, D, base = 12, 768, 5
    lens = [2, 77, 31, 1, 33, 64, 9]
    n = torch.tensor(lens, dtype=torch.int32)
    B = len(lens)
    cu = cu_of(n).cuda()
    qkv = _packed_qkv(n, D, 18, 0, base)
    rows = (base + cu[:B].long() + n.cuda().long() - 1)
    q = qkv[rows, :D].contiguous()
    poisoned = qkv.clone()
    poisoned[:, :D] = float("nan")
    out = torch.full((B + 2, D), 7.0, dtype=BF, device="cuda")
    hip.attention_lastq_varlen(q, poisoned, out, B, 77, Hh, cu, row_base=base)
    for b in range(B):
        r0, nk = base + int(cu[b]), lens[b]
        blk = qkv[r0:r0 + nk].float()
        k, v = blk[:, D:2 * D].reshape(nk, Hh, 64), blk[:, 2 * D:].reshape(nk, Hh, 64)
        s = torch.einsum("hd,khd->hk", q[b].float().reshape(Hh, 64), k)
        close(out[b], torch.einsum("hk,khd->hd", torch.softmax(s, -1), v).reshape(D), 2e-2, 2e-2)
    assert bool((out[B:] == 7.0).all())


@pytest.mark.parametrize("lens,causal", [([2, 77, 31, 1, 33], True), ([5, 9, 32, 1, 17], True), ([40, 64, 3], False), ([60, 62, 12, 4], True)])
def test_attention_backward_varlen(gpu_device, lens, causal):
    Hh, D = 12, 768
    n = torch.tensor(lens, dtype=torch.int32)
    B, total, pad = len(lens), int(n.sum()), 6
    cu = cu_of(n).cuda()
    qkv = (_packed_qkv(n, D, 9, pad).float() * 0.7).to(BF)
    dout = rnd(total + pad + 3, D, seed=10, dtype=BF)
    o = torch.zeros(total + pad + 3, D, dtype=BF, device="cuda")
    hip.attention_varlen(qkv, o, cu, B, int(n.max()), Hh, causal, pad_rows=pad)
    dqkv = torch.full_like(qkv, 7.0)
    hip.attention_bwd_varlen(qkv, o, dout, dqkv, cu, B, int(n.max()), Hh, causal, pad_rows=pad)
    for b in range(B):
        r0, L = int(cu[b]), lens[b]
        qf = qkv[r0:r0 + L].float().requires_grad_(True)
        q, k, v = (t.reshape(L, Hh, 64).transpose(0, 1) for t in qf.chunk(3, dim=-1))
        sc = q @ k.transpose(-1, -2)
        if causal:
            sc = sc + torch.full((L, L), float("-inf"), device="cuda").triu_(1)
        (torch.softmax(sc, -1) @ v).transpose(0, 1).reshape(L, D).backward(dout[r0:r0 + L].float())
        got = dqkv[r0:r0 + L].float()
        assert bool(torch.isfinite(got).all())
        err = (got - qf.grad).abs().max().item() / max(qf.grad.abs().max().item(), 1e-6)
        assert err < 3e-2, (b, err)
        if L > 1:
            assert F.cosine_similarity(got.flatten(), qf.grad.flatten(), dim=0).item() > 0.999, b
    assert bool((dqkv[total:total + pad] == 0).all()) and bool((dqkv[total + pad:] == 7.0).all())
    # with the per-caption token sums of dqkv (in_proj bias-gradient partials): same dqkv, sums over each caption's live rows
    d2 = torch.full_like(qkv, 7.0)
    part = torch.full((B + 1, 3 * D), float("nan"), dtype=torch.float32, device="cuda")
    hip.attention_bwd_varlen(qkv, o, dout, d2, cu, B, int(n.max()), Hh, causal, pad_rows=pad, colsum_part=part[:B])
    assert torch.equal(d2, dqkv) and bool(torch.isnan(part[B:]).all())
    for b in range(B):
        r0, L = int(cu[b]), lens[b]
        want = dqkv[r0:r0 + L].float().sum(0)
        bound = dqkv[r0:r0 + L].float().abs().sum(0) * 2.0 ** -8 + 1e-6
        assert bool(((part[b] - want).abs() <= bound).all()), b


def test_embedding_backward_packed(gpu_device):
    B, L, C, V = 37, 77, 768, 49408
    tok = edge_tokens(B, seed=8).cuda()
    n = lengths(tok.cpu())
    cu = cu_of(n).cuda()
    total = int(n.sum())
    dx = rnd(total + 4, C, seed=12)
    demb = torch.zeros(V, C, device="cuda")
    dpos = torch.full((L, C), 7.0, device="cuda")
    hip.embed_tokens_bwd_packed(tok, dx[:total], cu, demb, dpos)
    bidx = torch.repeat_interleave(torch.arange(B), n.long())
    lidx = torch.cat([torch.arange(int(k)) for k in n])
    ref_e = torch.zeros(V, C, device="cuda").index_add_(0, tok[bidx, lidx], dx[:total])
    ref_p = torch.zeros(L, C, device="cuda").index_add_(0, lidx.cuda(), dx[:total])
    close(demb, ref_e, 1e-5, 1e-5)
    close(dpos, ref_p, 1e-5, 1e-5)
    dpos2 = torch.empty_like(dpos)
    hip.embed_tokens_bwd_packed(tok, dx[:total], cu, torch.zeros_like(demb), dpos2)
    assert torch.equal(dpos, dpos2)                                  # fixed summation order


# ---------------------------------------------------------------------------------------------------------------- engine
_MODELS = {}


def _model(name):
    from conftest import synth_sd
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    if name not in _MODELS:
        m = get_clip_model(named_config(name))
        m.load_state_dict(synth_sd(name), strict=True)
        _MODELS[name] = m.cuda().eval()
    return _MODELS[name]


@pytest.mark.parametrize("B", [6, 512])
def test_engine_packed_text_equals_full_rows(gpu_device, monkeypatch, B):
    """encode_text / forward logits / contrastive loss with packed captions against the full-row path of the same engine
    (MSCLIP_TEXT_PACK=0) on a ragged batch with every edge caption: the two differ only in which GEMM tile / attention wave a
    row lands in (bf16-operand noise); B = 512 runs the LayerNorm fold over the packed segment and the side streams."""
    m = _model("b32-yfcc-msclips")
    eng = m.engine()
    img = synth.synth_images(B, seed=71).cuda()
    tok = edge_tokens(B, seed=72).cuda()
    set_opt(monkeypatch, eng, text_pack=False)
    w = eng.run(img, tok)
    assert not w["packed"] and w["Mt"] == B * 77
    f0i, f0t = w["fv"].clone(), w["ft"].clone()
    l0 = eng.forward_loss(img, tok, gather=False).item()
    lg0 = eng.forward_logits(img, tok, gather=False).clone()
    set_opt(monkeypatch, eng, text_pack=True)
    w = eng.run(img, tok)
    n = lengths(tok.cpu())
    assert w["packed"] and eng._live_text_rows(w) == int(n.sum()) and w["Lmax"] == 77 and torch.equal(w["len"].cpu(), n)
    assert w["dyn"] == (B == 512)                           # batch 512: the row count stays on the device (engine.dynamic_rows)
    assert w["Mt"] % 256 == 0 if B == 512 else w["Mt"] == w["Mt_live"]
    if B == 512:                                            # ... and the device-side dims say what the host-read path computed
        d = w["dims"].cpu().tolist()
        tot = int(n.sum())
        pad = -(-tot // 256) * 256
        assert d[:7] == [tot, 77, pad, w["Mv"] + pad, pad - tot, w["Mv"] + tot, w["Mv"]]
    f1i, f1t = w["fv"].clone(), w["ft"].clone()
    assert (f1i - f0i).abs().max().item() <= 2e-3 and (f1t - f0t).abs().max().item() <= 2e-3
    assert F.cosine_similarity(f1t, f0t, dim=-1).min().item() >= 0.99995
    assert (m.encode_text(tok) - f0t).abs().max().item() <= 2e-3              # text-only call (no side streams)
    assert abs(eng.forward_loss(img, tok, gather=False).item() - l0) <= 2e-3 * max(1.0, abs(l0))
    assert (eng.forward_logits(img, tok, gather=False) - lg0).abs().max().item() <= 0.05
    for _ in range(2):                                                          # bitwise repeatable
        w2 = eng.run(img, tok)
        assert torch.equal(w2["ft"], f1t) and torch.equal(w2["fv"], f1i)


@pytest.mark.parametrize("lens", ["ragged", "edge", "short"])
def test_device_side_row_counts_are_bitwise_the_host_sized_path(gpu_device, monkeypatch, lens):
    """engine.dynamic_rows: the packed row count stays on the device (msclip_text_lengths' dims -> msclip_gemm_desc.M_dev and the
    m_dev / dims arguments; every launch is sized for the upper bound) against the round-5 path where the host reads the total
    and sizes every launch exactly (dynamic_rows=False): the same tiles run the same arithmetic, so features, loss and logits
    are BITWISE equal -- for ragged captions, the edge captions, and a batch so short that most of the bound is empty.  Then
    with stale garbage behind the live rows (a longer batch ran before): still bitwise."""
    m = _model("b32-yfcc-msclips")
    eng = m.engine()
    B = 512
    img = synth.synth_images(B, seed=171).cuda()
    tok = {"ragged": lambda: synth.synth_tokens(B, seed=172), "edge": lambda: edge_tokens(B, seed=173),
           "short": lambda: synth.synth_tokens(B, seed=174, min_len=7, max_len=9)}[lens]().cuda()   # (>= 4096 rows: the host-sized path pads and folds too)
    long_tok = synth.synth_tokens(B, seed=175, min_len=70, max_len=75).cuda()
    set_opt(monkeypatch, eng, dynamic_rows=False, plan=False)
    w = eng.run(img, tok)
    assert w["packed"] and not w["dyn"]
    f0i, f0t = w["fv"].clone(), w["ft"].clone()
    l0 = eng.forward_loss(img, tok, gather=False).clone()
    lg0 = eng.forward_logits(img, tok, gather=False).clone()
    for plan in (False, True):
        set_opt(monkeypatch, eng, dynamic_rows=True, plan=plan)
        eng.run(img, long_tok)                               # leaves other values in every row behind this batch's live rows
        for _ in range(3 if plan else 1):                    # (plan: the recording pass, then replays)
            w = eng.run(img, tok)
            assert w["packed"] and w["dyn"]
            assert torch.equal(w["fv"], f0i) and torch.equal(w["ft"], f0t)
        assert torch.equal(eng.forward_loss(img, tok, gather=False), l0)
        assert torch.equal(eng.forward_logits(img, tok, gather=False), lg0)
        assert torch.equal(m.encode_text(tok), f0t) or (m.encode_text(tok) - f0t).abs().max().item() <= 2e-3   # text-only call: other tile map


def test_engine_packed_text_against_oracle(gpu_device):
    """Packed captions against the fp32 oracle (which computes all 77 rows like the reference): text features of the edge
    captions and a ragged batch, stated tolerance (DESIGN s2)."""
    from conftest import synth_sd
    from oracle import msclip_oracle as O
    name = "b32-yfcc-msclips"
    m = _model(name)
    tok = edge_tokens(24, seed=5)
    with torch.no_grad():
        ref = O.encode_text(tok, synth_sd(name), O.arch_b32())
    got = m.encode_text(tok.cuda()).cpu()
    assert m.engine()._ws[(0, 24)]["packed"]
    assert (got - ref).abs().max().item() <= 5e-3
    assert F.cosine_similarity(got, ref, dim=-1).min().item() >= 0.9999


def test_text_block_taps_on_live_rows(gpu_device, monkeypatch):
    """Text-block taps of the packed path against the full-row path's, on the live rows (the others are zero and flagged by
    taps["text_lengths"])."""
    m = _model("b32-yfcc-msclips")
    eng = m.engine()
    B = 8
    img = synth.synth_images(B, seed=73).cuda()
    tok = edge_tokens(B, seed=74).cuda()
    set_opt(monkeypatch, eng, text_pack=False)
    t0 = {}
    eng.run(img, tok, taps=t0)
    set_opt(monkeypatch, eng, text_pack=True)
    t1 = {}
    eng.run(img, tok, taps=t1)
    n = t1["text_lengths"].long()
    assert torch.equal(n.cpu(), lengths(tok.cpu()).long()) and "text_lengths" not in t0
    live = (torch.arange(77, device="cuda")[None, :] < n[:, None])
    keys = [k for k in t0 if k.startswith("tblock")]
    assert len(keys) >= eng.n_layers - 1
    for k in keys:
        a, b = t0[k], t1[k]
        assert a.shape == b.shape
        assert bool((b[~live] == 0).all())
        err = ((a - b).abs() * live[:, :, None]).max().item() / max(a[live].abs().max().item(), 1e-3)
        assert err <= 1.5e-2, (k, err)
    for k in t0:
        if k.startswith("vblock") or k.startswith("adapter"):
            assert (t0[k] - t1[k]).abs().max().item() <= 1.5e-2 * max(t0[k].abs().max().item(), 1e-3), k


def test_training_gradients_packed_equal_full_rows(gpu_device, monkeypatch):
    """Every gradient of the training step with packed captions against the same step over all 77 rows per caption
    (MSCLIP_TEXT_PACK=0; that path is pinned to the reference's autograd by tests/test_gpu_train.py, which itself runs packed
    by default): a row behind the EOT position has zero upstream gradient, so the two agree to bf16-operand noise -- on a ragged
    batch with the edge captions (EOT at 0 / 1, 77 live rows, a second EOT id)."""
    from msclip_amd import train
    m = _model("b32-yfcc-msclips")
    B = 12
    img = synth.synth_images(B, seed=81).cuda()
    tok = edge_tokens(B, seed=82).cuda()
    out = {}
    for mode in ("0", "1"):
        set_opt(monkeypatch, m, text_pack=(mode == "1"))
        ts = train.TrainStep(m, lr=1e-4)
        loss = ts.forward(img, tok if mode == "0" else m.stage_captions(tok))      # (packed: through a staged batch)
        assert (ts.saved["cap"] is not None) == (mode == "1")
        out[mode] = (loss.item(), {k: v.float().clone() for k, v in ts.backward().items()})
    (l0, g0), (l1, g1) = out["0"], out["1"]
    assert abs(l0 - l1) <= 2e-3 * max(1.0, abs(l0))
    assert sorted(g0) == sorted(g1) and len(g0) == 325
    worst = {}
    for k in g0:
        a, b = g0[k].flatten(), g1[k].flatten()
        scale = max(a.abs().max().item(), 1e-12)
        worst[k] = (a - b).abs().max().item() / scale
        cos = F.cosine_similarity(a, b, dim=0).item() if a.numel() > 1 else 1.0
        assert worst[k] <= 6e-2 and cos >= 0.995, (k, worst[k], cos)
    print("packed vs full-row gradients: worst", sorted(worst.items(), key=lambda kv: -kv[1])[:4],
          "median", float(np.median(list(worst.values()))))
    assert float(np.median(list(worst.values()))) <= 1e-2
    # the positional embedding's gradient rows that no caption reaches are exactly zero in both
    lmax = int(lengths(tok.cpu()).max())
    assert lmax == 77 or bool((g1["positional_embedding"][lmax:] == 0).all())


def test_captions_staged_on_a_prefetch_stream(gpu_device, monkeypatch):
    """a batch staged on ANOTHER stream than the one that consumes it (an input pipeline's prefetch stage,
    Engine.stage_captions) -- its device tensors belong to the staging stream's pool, so the engine has to tell the allocator
    about the consuming streams (record_stream) or a block freed with the batch could be handed to the next stage_captions
    while kernels still read it.  Stage on a side stream, drop the batch right after the call, stage the next ones on the same
    stream immediately: features equal the plain-tensor call, for the host-sized path (which reads the staged cu / eot) and the
    device-side one."""
    m = _model("b32-yfcc-msclips")
    eng = m.engine()
    B = 512
    img = synth.synth_images(B, seed=301).cuda()
    toks = [synth.synth_tokens(B, seed=310 + i, min_len=2 + i, max_len=30 + 9 * i).cuda() for i in range(4)]
    ref = [eng.run(img, t)["ft"].clone() for t in toks]
    prefetch = torch.cuda.Stream()
    for dyn in (False, True):
        set_opt(monkeypatch, eng, dynamic_rows=dyn, plan=dyn)
        for rep in range(3):
            for t, want in zip(toks, ref):
                prefetch.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(prefetch):
                    cap = eng.stage_captions(t.clone())              # (the clone is the prefetch stage's own buffer)
                torch.cuda.current_stream().wait_stream(prefetch)
                w = eng.run(img, cap)
                del cap                                              # blocks go back to the prefetch stream's pool at once
                with torch.cuda.stream(prefetch):                    # ... and the next stage grabs memory there while the step runs
                    junk = [torch.full((B + 2,), -1, dtype=torch.int32, device="cuda") for _ in range(8)]
                assert torch.equal(w["ft"], want), (dyn, rep)
                del junk
