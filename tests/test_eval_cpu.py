"""Caller-side pieces that run without a GPU: tokenizer (vs ids captured from the reference tokenizer, and vs a
brute-force BPE on a synthetic merge table), image preprocessing, accuracy, checkpoint round trip with aliases."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, synth_sd
from msclip_amd import checkpoint, zeroshot
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
from msclip_amd.tokenizer import SimpleTokenizer, find_vocab


def _has_vocab():
    try:
        find_vocab()
        return True
    except FileNotFoundError:
        return False


@pytest.mark.skipif(not _has_vocab(), reason="CLIP merges file (reference data) not available")
def test_tokenizer_matches_reference_ids():
    g = json.load(open(os.path.join(GOLDEN, "tokenizer.json")))
    tok = SimpleTokenizer()
    assert tok.get_vocab_size() == 49408 and tok.get_sot_token() == g["sot"] and tok.get_eot_token() == g["eot"]
    assert tok(g["prompts"]).tolist() == g["ids"]
    assert tok("x").shape == (1, 77) and tok(["a", "b"], context_length=8).shape == (2, 8)


def test_tokenizer_synthetic_merges(tmp_path):
    """Greedy lowest-rank merging against an independent brute-force implementation on a tiny merge table."""
    merges = [("t", "h"), ("th", "e</w>"), ("a", "n"), ("an", "d</w>"), ("c", "a"), ("ca", "t</w>"), ("t", "he</w>")]
    path = tmp_path / "v.txt.gz"
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    tok = SimpleTokenizer(str(path), vocab_size=512 + len(merges) + 2)

    def brute(word):
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        rank = {m: i for i, m in enumerate(merges)}
        while True:
            cands = [(rank[p], p) for p in zip(parts, parts[1:]) if p in rank]
            if not cands:
                return parts
            _, (a, b) = min(cands)
            out, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and (parts[i], parts[i + 1]) == (a, b):
                    out.append(a + b); i += 2
                else:
                    out.append(parts[i]); i += 1
            parts = out

    for w in ["the", "and", "cat", "that", "thethe", "a", "tha"]:
        assert [tok.decoder[i] for i in tok.encode(w)] == brute(w), w
    ids = tok.tokenize("the cat", context_length=6)[0].tolist()
    assert ids[0] == tok.sot_token and ids[3] == tok.eot_token and ids[4:] == [0, 0]
    assert tok.decode(tok.encode("the cat and")) == "the cat and "


def test_preprocess_and_accuracy(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    x = zeroshot.preprocess(Image.fromarray(arr))
    assert x.shape == (3, 224, 224) and x.dtype == torch.float32
    # identity geometry: a 224x224 image is only scaled and normalised
    sq = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    y = zeroshot.preprocess(Image.fromarray(sq))
    ref = (sq.astype(np.float32) / 255.0 - np.array(zeroshot.IMAGENET_MEAN, np.float32)) / np.array(zeroshot.IMAGENET_STD, np.float32)
    assert np.abs(y.numpy() - ref.transpose(2, 0, 1)).max() < 1e-6
    out = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1], [0.2, 0.3, 0.5]])
    assert zeroshot.accuracy(out, torch.tensor([1, 0, 0]), (1, 2)) == [pytest.approx(200 / 3), pytest.approx(200 / 3)]
    for c in ("n02", "n01"):
        os.makedirs(tmp_path / "val" / c)
        Image.fromarray(arr).save(tmp_path / "val" / c / "a.png")
    classes, items = zeroshot.image_folder(str(tmp_path / "val"))
    assert classes == ["n01", "n02"] and [c for _, c in items] == [0, 1]


def test_checkpoint_roundtrip_and_alias_check(tmp_path):
    name = "b32-yfcc-msclips"
    model = get_clip_model(named_config(name))
    sd = synth_sd(name)
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}, "epoch": 3}, tmp_path / "ck.pth")   # trainer format
    checkpoint.load_pretrained(model, str(tmp_path / "ck.pth"))
    assert checkpoint.check_aliases(model) == 88
    checkpoint.save_model(model, str(tmp_path / "bare.pth"))                                                   # eval format
    back = torch.load(tmp_path / "bare.pth")
    assert list(back) == list(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    bad = dict(sd)
    bad["transformer.resblocks.4.mlp.c_fc.bias"] = bad["transformer.resblocks.4.mlp.c_fc.bias"] + 1
    torch.save(bad, tmp_path / "bad.pth")
    with pytest.raises(RuntimeError, match="disagrees"):
        checkpoint.load_pretrained(get_clip_model(named_config(name)), str(tmp_path / "bad.pth"))


def test_text_repair_follows_ftfy_examples():
    """msclip_amd.textfix.fix_text -- the stand-in for `ftfy.fix_text` (reference simple_tokenizer.py:54-57) where the package is
    absent -- on the known-answer examples of ftfy's own README / docs (mojibake incl. a run embedded in healthy text and a doubly
    mangled apostrophe, curly quotes, full-width forms, ligatures, line breaks, terminal escapes, NFC), on text it must leave alone,
    and through the tokenizer: the repaired caption tokenises like its clean spelling."""
    from msclip_amd import textfix
    from msclip_amd.tokenizer import SimpleTokenizer, basic_clean
    e_acute = "é"
    cases = [("âœ” No problems", "✔ No problems"),
             ("l’humanitÃ©", "l'humanit" + e_acute),
             ("ＬＯＵＤ　ＮＯＩＳＥＳ", "LOUD NOISES"),
             ("ﬂuﬃest", "fluffiest"),
             ("“quoted” ‘single’", "\"quoted\" 'single'"),
             ("doesnÃ¢â‚¬â„¢t", "doesn't"),
             ("line one\r\nline two three", "line one\nline two\nthree"),
             ("\x1b[36;44mblue\x1b[0m", "blue"),
             ("é", e_acute),
             # left alone: healthy Latin-1 text (its bytes are not UTF-8), capitals that only look like a lead byte, plain ASCII
             ("a photo of a caf" + e_acute, "a photo of a caf" + e_acute),
             ("SÃO PAULO", "SÃO PAULO"),
             ("a photo of a tench.", "a photo of a tench.")]
    for raw, want in cases:
        assert textfix.fix_text(raw) == want, (ascii(raw), ascii(textfix.fix_text(raw)), ascii(want))
        assert textfix.fix_text(want) == want                        # idempotent on its own output
    tok = SimpleTokenizer()
    for raw, want in cases[:6]:
        assert basic_clean(raw) == basic_clean(want)
        assert torch.equal(tok([raw]), tok([want])), ascii(raw)
