"""End-to-end parity of the HIP path (bf16 operands, fp32 accumulation/residual) on a real MI355X:
against the golden vectors captured from the reference, against the CPU oracle on other seeds / ragged batches,
and through size-independent properties at the benchmark's full batch.

Stated tolerance for bf16 vs the fp32 reference: unit-norm features max-abs <= 5e-3 and
cosine >= 0.9999; logits (T = 1/0.07) max-abs <= 0.05; the reference's own CPU bf16-autocast run differs from
its fp32 run by 1.2e-3 / cos 0.99998."""
import numpy as np
import pytest
import torch

from conftest import golden, synth_sd
from msclip_amd import hip, synth
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
from oracle import msclip_oracle as O

pytestmark = pytest.mark.gpu
FEAT_TOL, COS_TOL, LOGIT_TOL = 5e-3, 0.9999, 0.05
_MODELS = {}


def model_for(name):
    if name not in _MODELS:
        m = get_clip_model(named_config(name))
        m.load_state_dict(synth_sd(name), strict=True)
        _MODELS[name] = m.cuda().eval()
    return _MODELS[name]


def check_feats(got, ref):
    got, ref = got.float().cpu(), torch.as_tensor(ref).float()
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=-1).min().item()
    assert err <= FEAT_TOL and cos >= COS_TOL, f"max-abs {err:.3e}, min cos {cos:.6f}"
    return err, cos


@pytest.mark.parametrize("name", ["b32-yfcc-msclips", "b16-yfcc-msclips"])
def test_against_reference_golden(gpu_device, name):
    g = golden(name)
    m = model_for(name)
    b = int(g["batch"])
    img = synth.synth_images(b, seed=int(g["seed"])).cuda()
    tok = synth.synth_tokens(b, seed=int(g["seed"]) + 1).cuda()
    check_feats(m.encode_image(img), g["image_features"])
    check_feats(m.encode_text(tok), g["text_features"])
    logits = m(img, tok).cpu()
    assert logits.shape == (b, b)
    assert np.abs(logits.numpy() - g["logits"]).max() <= LOGIT_TOL
    raw = m.encode_image(img, norm=False).cpu().numpy()
    ref = g["image_features_raw"]
    assert np.abs(raw - ref).max() <= 2e-2 * np.abs(ref).max()
    # the native library really is what ran
    with open("/proc/self/maps") as f:
        assert "libmsclip_hip.so" in f.read()


def test_against_oracle_ragged_batches_and_loss(gpu_device):
    name, arch = "b32-yfcc-msclips", O.arch_b32()
    m, sd = model_for(name), synth_sd(name)
    img = synth.synth_images(3, seed=21)
    tok = synth.synth_tokens(5, seed=22, min_len=1, max_len=75)
    tok[0, :] = 0
    tok[0, 0], tok[0, 1] = 49406, 49407                              # shortest caption: EOT at position 1
    with torch.no_grad():
        ri, rt = O.encode_image(img, sd, arch), O.encode_text(tok, sd, arch)
    check_feats(m.encode_image(img.cuda()), ri)
    check_feats(m.encode_text(tok.cuda()), rt)
    # joint run (towers batched through the shared layers) must agree with the separate runs
    with torch.no_grad():
        ri5 = O.encode_image(synth.synth_images(5, seed=23), sd, arch)
    w = m.engine().run(synth.synth_images(5, seed=23).cuda(), tok.cuda())
    check_feats(w["fv"], ri5)
    check_feats(w["ft"], rt)
    lg = m(synth.synth_images(5, seed=23).cuda(), tok.cuda()).cpu()
    ref_lg = O.clip_logits(ri5, rt, sd["logit_scale"])
    assert (lg - ref_lg).abs().max() <= LOGIT_TOL
    loss = m.contrastive_loss(synth.synth_images(5, seed=23).cuda(), tok.cuda()).item()
    assert abs(loss - O.contrastive_loss(ref_lg).item()) <= 2e-2
    assert abs(loss - O.contrastive_loss(lg).item()) <= 2e-3          # fused/sharded form == full-matrix form


def test_batch_invariance_and_determinism(gpu_device):
    m = model_for("b32-yfcc-msclips")
    img = synth.synth_images(6, seed=31).cuda()
    a = m.encode_image(img)
    b = m.encode_image(img)
    assert torch.equal(a, b)                                          # no atomics / races: bitwise repeatable
    c = m.encode_image(img[2:3])
    assert (a[2:3] - c).abs().max().item() <= 1e-5                    # per-sample result independent of the batch


@pytest.mark.parametrize("name", ["b32-yfcc-msclips", "b16-yfcc-msclips"])
def test_fused_front_paths_against_the_conv_by_conv_chain(gpu_device, monkeypatch, name):
    """The fused stem pass / fused first bottleneck and the kernel-by-kernel chain they replace give the same image
    features (same bf16 roundings of every tensor that exists in both; the fused block keeps the shortcut in fp32)."""
    m = model_for(name)
    img = synth.synth_images(5, seed=37).cuda()
    fused = m.encode_image(img)
    monkeypatch.setenv("MSCLIP_BLOCK_UNFUSED", "1")
    no_block = m.encode_image(img)
    monkeypatch.setenv("MSCLIP_FRONT_UNFUSED", "1")
    chain = m.encode_image(img)
    monkeypatch.setenv("MSCLIP_FRONT_4WAVE", "1")
    monkeypatch.setenv("MSCLIP_FRONT_UNFUSED", "0")
    monkeypatch.setenv("MSCLIP_BLOCK_UNFUSED", "1")
    four_wave = m.encode_image(img)
    for other in (no_block, chain, four_wave):
        assert (fused - other).abs().max().item() <= 2e-3
        assert torch.nn.functional.cosine_similarity(fused, other, dim=-1).min().item() >= 0.99995


def test_full_bench_batch_properties(gpu_device):
    """BASELINE config C2 (B = 512): finite, unit-norm, logits = scaled cosine, loss identity, rows match a small run."""
    m = model_for("b32-yfcc-msclips")
    B = 512
    img = synth.synth_images(B, seed=41).cuda()
    tok = synth.synth_tokens(B, seed=42).cuda()
    w = m.engine().run(img, tok)
    fi, ft = w["fv"].clone(), w["ft"].clone()
    assert torch.isfinite(fi).all() and torch.isfinite(ft).all()
    assert (fi.norm(dim=-1) - 1).abs().max().item() < 1e-4 and (ft.norm(dim=-1) - 1).abs().max().item() < 1e-4
    lg = m(img, tok)
    ref = m.engine().logit_scale_exp * fi @ ft.t()
    assert (lg - ref).abs().max().item() <= 0.05
    loss = m.contrastive_loss(img, tok).item()
    lab = torch.arange(B, device="cuda")
    ce = 0.5 * (torch.nn.functional.cross_entropy(lg, lab) + torch.nn.functional.cross_entropy(lg.t(), lab)).item()
    assert abs(loss - ce) <= 2e-3
    small_i = m.encode_image(img[100:104])
    small_t = m.encode_text(tok[300:303])
    # the small run goes through other tile configurations (fp32 sums associate differently -> bf16 rounding flips)
    assert (small_i - fi[100:104]).abs().max().item() <= 2e-3
    assert (small_t - ft[300:303]).abs().max().item() <= 2e-3
    # and the full-size path (large-tile kernels everywhere) is pinned to the oracle on a few of its samples
    sd, arch = synth_sd("b32-yfcc-msclips"), O.arch_b32()
    with torch.no_grad():
        ri, rt = O.encode_image(img[[0, 255, 511]].cpu(), sd, arch), O.encode_text(tok[[1, 256, 510]].cpu(), sd, arch)
    check_feats(fi[[0, 255, 511]], ri)
    check_feats(ft[[1, 256, 510]], rt)
    # single-tower calls at the same size (other row counts -> other tile grids)
    check_feats(m.encode_image(img)[[0, 255, 511]], ri)
    check_feats(m.encode_text(tok)[[1, 256, 510]], rt)


def test_full_batch_b16_against_oracle(gpu_device):
    """BASELINE config C3 (ViT-B/16, B = 256: 197-token attention, 14 x 14 adapters) against the oracle on a few samples."""
    name = "b16-yfcc-msclips"
    m, sd, arch = model_for(name), synth_sd(name), O.arch_b16()
    B = 256
    img = synth.synth_images(B, seed=43).cuda()
    tok = synth.synth_tokens(B, seed=44).cuda()
    w = m.engine().run(img, tok)
    fi, ft = w["fv"].clone(), w["ft"].clone()
    assert torch.isfinite(fi).all() and torch.isfinite(ft).all()
    with torch.no_grad():
        check_feats(fi[[0, 128, 255]], O.encode_image(img[[0, 128, 255]].cpu(), sd, arch))
        check_feats(ft[[2, 100, 254]], O.encode_text(tok[[2, 100, 254]].cpu(), sd, arch))


def test_inputs_validated(gpu_device):
    m = model_for("b32-yfcc-msclips")
    with pytest.raises(ValueError):
        m.encode_image(torch.zeros(1, 3, 128, 128, device="cuda"))
    with pytest.raises(ValueError):
        m.encode_text(torch.zeros(1, 60, dtype=torch.long, device="cuda"))
    with pytest.raises(hip.HipUnavailable):
        m.encode_text(torch.zeros(1, 77, dtype=torch.long))


def test_zeroshot_pipeline_against_oracle(gpu_device, tmp_path):
    """BASELINE config C1 plumbing (zero-shot driver) on generated images: classifier columns, 100*f@W logits and
    the top-1 decisions must follow the oracle's (reference tools/zero_shot.py:122-134, 265-266)."""
    import zlib
    from PIL import Image
    from msclip_amd import zeroshot
    name, arch = "b32-yfcc-msclips", O.arch_b32()
    m, sd = model_for(name), synth_sd(name)

    class FakeTok:                                       # deterministic stand-in for the BPE (vocab file not on this box)
        def __call__(self, texts, context_length=77):
            if isinstance(texts, str):
                texts = [texts]
            out = torch.zeros(len(texts), context_length, dtype=torch.long)
            for i, t in enumerate(texts):
                r = np.random.default_rng(zlib.crc32(t.encode()))
                n = int(r.integers(3, 20))
                out[i, 0], out[i, 1:1 + n], out[i, 1 + n] = 49406, torch.from_numpy(r.integers(1, 49000, n)), 49407
            return out

    tok = FakeTok()
    classes = ["tench", "goldfish", "shark"]
    templates = ["a photo of a {}.", "a bad photo of a {}.", "art of the {}.", "itap of a {}."]
    W = zeroshot.zeroshot_classifier(m, tok, classes, templates, classes_per_batch=2)
    with torch.no_grad():
        Wref = O.zeroshot_classifier([tok([t.format(c) for t in templates]) for c in classes], sd, arch)
    assert W.shape == (512, 3) and (W.float().cpu() - Wref).abs().max().item() <= FEAT_TOL
    rng = np.random.default_rng(5)
    for ci, c in enumerate(["n01", "n02", "n03"]):
        (tmp_path / "val" / c).mkdir(parents=True)
        for k in range(3):
            Image.fromarray(rng.integers(0, 256, (240 + 10 * k, 260, 3), dtype=np.uint8)).save(tmp_path / "val" / c / f"{k}.png")
    res = zeroshot.evaluate(m, tok, str(tmp_path / "val"), classes, templates, batch_size=4, log=lambda s: None)
    assert res["n"] == 9 and 0.0 <= res["top1"] <= 100.0
    _, items = zeroshot.image_folder(str(tmp_path / "val"))
    x = torch.stack([zeroshot.preprocess(Image.open(p)) for p, _ in items])
    with torch.no_grad():
        ref_logits = 100.0 * O.encode_image(x, sd, arch) @ Wref
    got_logits = 100.0 * m.encode_image(x.cuda()).float().cpu() @ W.float().cpu()
    assert (got_logits - ref_logits).abs().max().item() <= 0.3        # stated zero-shot logit tolerance (x100 scale)
    y = torch.tensor([c for _, c in items])
    assert abs(res["top1"] - zeroshot.accuracy(ref_logits, y)[0]) <= 100.0 / 9 + 1e-6


def test_hipgraph_replay_matches_eager(gpu_device):
    m = model_for("b32-yfcc-msclips")
    eng = m.engine()
    img = synth.synth_images(4, seed=61).cuda()
    tok = synth.synth_tokens(4, seed=62).cuda()
    w = eng.run(img, tok)
    fi, ft = w["fv"].clone(), w["ft"].clone()
    replay = eng.graph(4, 4)
    w2 = replay(img, tok)
    assert torch.equal(w2["fv"], fi) and torch.equal(w2["ft"], ft)          # same kernels, same order: bitwise equal
    img2 = synth.synth_images(4, seed=63).cuda()
    w3 = replay(img2, tok)
    assert torch.equal(w3["fv"], eng.run(img2, tok)["fv"])
    only_txt = eng.graph(0, 80)
    t80 = synth.synth_tokens(80, seed=64).cuda()
    assert torch.equal(only_txt(tok=t80)["ft"], eng.run(tok=t80)["ft"])
