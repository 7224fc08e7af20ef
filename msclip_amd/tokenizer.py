"""CLIP byte-pair tokenizer (caller side of the text tower).

Behaviour of the reference's `SimpleTokenizer` (lib/dataset/languages/simple_tokenizer.py:66-169,225):
text -> html-unescape twice, collapse whitespace, lower-case -> regex pre-tokens -> UTF-8 bytes mapped to printable
code points -> greedy lowest-rank pair merges with an end-of-word marker -> ids; `tokenize` wraps with SOT / EOT,
zero-pads to `context_length` and truncates longer sequences (the reference truncates silently, :160-163).

The merge table is DATA: the 48 894 merges the 49 408-entry vocabulary uses ship as
msclip_amd/data/clip_bpe_merges.txt.gz (extracted by tools/make_data.py; the released checkpoints' token ids are
defined by it).  `bpe_path=` / MSCLIP_BPE_VOCAB may point at another table in the same format, e.g. the reference's
full `bpe_simple_vocab_16e6.txt.gz`.

Text cleaning: the reference calls `ftfy.fix_text` first (simple_tokenizer.py:54-57).  It is used when importable;
otherwise (ftfy is not installed in this image) msclip_amd.textfix.fix_text runs: a restatement of ftfy's default pipeline
(mojibake repair, curly quotes, ligatures, full-width forms, line breaks, control characters, NFC; round 6 -- rounds 1-5 only
NFC-normalised).  Ids are identical to the reference's on text ftfy leaves alone (all ASCII, and already-normalised Unicode --
pinned by tests/golden/tokenizer.json, which has non-ASCII cases); on text ftfy rewrites they follow the restatement, which is
pinned to ftfy's README examples only (parity unpinned against the package itself).
"""
import gzip
import html
import logging
import os
import unicodedata

import regex
import torch

from . import textfix as _textfix

_PRETOKEN = regex.compile(
    r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
_EOW = "</w>"
_PACKAGED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "clip_bpe_merges.txt.gz")
_CANDIDATE_PATHS = (_PACKAGED, "bpe_simple_vocab_16e6.txt.gz", "lib/dataset/languages/bpe_simple_vocab_16e6.txt.gz")

try:
    import ftfy as _ftfy
except ImportError:
    _ftfy = None
_warned = []


def basic_clean(text):
    """simple_tokenizer.py:54-57: ftfy.fix_text, html.unescape twice, strip."""
    if _ftfy is not None:
        text = _ftfy.fix_text(text)
    else:
        if not text.isascii() and not _warned:
            _warned.append(1)
            logging.getLogger(__name__).info("ftfy is not installed: non-ASCII captions are repaired by msclip_amd.textfix "
                                             "(a restatement of ftfy.fix_text's default pipeline)")
        text = _textfix.fix_text(text)
    return html.unescape(html.unescape(text)).strip()


def find_vocab(path=None):
    for p in ([path] if path else []) + [os.environ.get("MSCLIP_BPE_VOCAB", "")] + list(_CANDIDATE_PATHS):
        if p and os.path.isfile(p):
            return p
    raise FileNotFoundError("CLIP BPE merges file not found: pass bpe_path= or set MSCLIP_BPE_VOCAB "
                            "(the reference ships it as lib/dataset/languages/bpe_simple_vocab_16e6.txt.gz)")


def _byte_alphabet():
    """256 printable stand-ins for the byte values (printable bytes map to themselves, the rest to U+0100...)."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class SimpleTokenizer:
    def __init__(self, bpe_path=None, vocab_size=49408):
        self.byte_map = _byte_alphabet()
        self.byte_unmap = {c: b for b, c in self.byte_map.items()}
        n_merges = vocab_size - 256 - 256 - 2                      # 48894 merges for the 49408-entry vocabulary
        with gzip.open(find_vocab(bpe_path), "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")[1:1 + n_merges]           # line 0 is a version header
        merges = [tuple(line.split()) for line in lines]
        # vocabulary order: byte symbols in byte-value-class order, the same with the end-of-word marker, merges, specials
        base = [self.byte_map[b] for b in sorted(self.byte_map, key=lambda b: (self.byte_map[b] != chr(b), b))]
        symbols = base + [s + _EOW for s in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {s: i for i, s in enumerate(symbols)}
        self.decoder = {i: s for s, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self._memo = {"<|startoftext|>": ["<|startoftext|>"], "<|endoftext|>": ["<|endoftext|>"]}
        self.sot_token, self.eot_token = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]

    # ---- reference accessor names
    def get_vocab_size(self):
        return len(self.encoder)

    def get_eot_token(self):
        return self.eot_token

    def get_sot_token(self):
        return self.sot_token

    def _merge_word(self, token):
        """Repeatedly fuse the adjacent pair with the lowest merge rank (all its occurrences, left to right)."""
        hit = self._memo.get(token)
        if hit is not None:
            return hit
        parts = list(token[:-1]) + [token[-1] + _EOW]
        while len(parts) > 1:
            best, best_rank = None, None
            for pair in zip(parts, parts[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            fused, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    fused.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    fused.append(parts[i])
                    i += 1
            parts = fused
        self._memo[token] = parts
        return parts

    def encode(self, text):
        text = basic_clean(text)
        text = regex.sub(r"\s+", " ", text).strip().lower()
        ids = []
        for piece in _PRETOKEN.findall(text):
            mapped = "".join(self.byte_map[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[s] for s in self._merge_word(mapped))
        return ids

    def decode(self, ids):
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.byte_unmap[c] for c in text).decode("utf-8", errors="replace").replace(_EOW, " ")

    def tokenize(self, texts, context_length=77):
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = ([self.sot_token] + self.encode(t) + [self.eot_token])[:context_length]
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out

    __call__ = tokenize
