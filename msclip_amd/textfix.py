"""Text repair in front of the BPE tokenizer: what `ftfy.fix_text` does for the reference (lib/dataset/languages/
simple_tokenizer.py:54-57 calls it on every caption), restated from ftfy's published behaviour (v5 / v6 defaults) because the
package is a third-party dependency that is absent from this image and from /root/reference (no pinned version: the reference's
INSTALL.md lists `ftfy` without one).  PARITY UNPINNED against ftfy itself; pinned only by the known-answer examples of ftfy's own
README that tests/test_eval_cpu.py holds.  When `ftfy` IS importable the tokenizer calls it instead (tokenizer.basic_clean).

Steps, in ftfy's order, repeated until the text stops changing:
  * mojibake: runs of characters that are the cp1252 / latin-1 reading of a UTF-8 byte sequence (a lead byte 0xC2-0xF4 followed
    by its continuation bytes) are re-encoded and decoded as UTF-8 when that is valid -- the two-character reading of e-acute
    becomes e-acute again, the three-character reading of a curly quote becomes the quote; doubly mangled text too; a run that
    does not decode is left alone.  (ftfy additionally scores "badness" before it touches two-byte runs; this restatement
    repairs every run that decodes.)
  * C1 control characters that are cp1252 punctuation read as latin-1 -> the cp1252 character
  * Latin ligatures (fi-ligature -> fi ...), full-width / half-width forms -> their normal forms, curly quotes -> straight quotes
  * line breaks (CRLF, CR, U+2028, U+2029, NEL) -> LF; terminal escapes and non-printing control characters removed; BOM removed
  * NFC normalisation
"""
import re
import unicodedata

# ---- sloppy cp1252: the five bytes cp1252 leaves undefined map to the latin-1 control characters (as ftfy's codec does)
_CP1252 = {}
for _b in range(256):
    try:
        _CP1252[bytes([_b]).decode("cp1252")] = _b
    except UnicodeDecodeError:
        _CP1252[chr(_b)] = _b
for _b in range(256):                       # latin-1 readings of the same bytes (C1 controls, and everything cp1252 shares)
    _CP1252.setdefault(chr(_b), _b)

_LEAD = "".join(re.escape(c) for c, b in _CP1252.items() if 0xC2 <= b <= 0xF4)
_CONT = "".join(re.escape(c) for c, b in _CP1252.items() if 0x80 <= b <= 0xBF)
_MOJIBAKE = re.compile(f"[{_LEAD}][{_CONT}]{{1,3}}")

_LIGATURES = {0xFB00: "ff", 0xFB01: "fi", 0xFB02: "fl", 0xFB03: "ffi", 0xFB04: "ffl", 0xFB05: "ſt", 0xFB06: "st",
              0x0132: "IJ", 0x0133: "ij", 0x0149: "ʼn", 0x01F1: "DZ", 0x01F2: "Dz", 0x01F3: "dz",
              0x01C4: "DŽ", 0x01C5: "Dž", 0x01C6: "dž", 0x01C7: "LJ", 0x01C8: "Lj", 0x01C9: "lj",
              0x01CA: "NJ", 0x01CB: "Nj", 0x01CC: "nj"}
_QUOTES = {0x2018: "'", 0x2019: "'", 0x201A: "'", 0x201B: "'", 0x02BC: "'", 0x201C: '"', 0x201D: '"', 0x201E: '"', 0x201F: '"'}
# full-width ASCII variants, the ideographic space, half-width katakana / hangul etc.: whatever NFKC folds inside these blocks
_WIDTH = {c: unicodedata.normalize("NFKC", chr(c)) for c in list(range(0xFF01, 0xFFEF)) + [0x3000]
          if unicodedata.normalize("NFKC", chr(c)) != chr(c)}
_TRANSLATE = {**_LIGATURES, **_QUOTES, **_WIDTH}
_LINE_BREAKS = re.compile("\r\n|\r| | |\x85")
_ANSI = re.compile("\x1b\\[[0-?]*[ -/]*[@-~]")
# control characters ftfy removes: C0 except tab / LF / the whitespace it keeps, DEL, deprecated format characters, BOM,
# interlinear annotation marks
_CONTROL = dict.fromkeys(list(range(0x00, 0x09)) + [0x0B] + list(range(0x0E, 0x20)) + [0x7F] +
                         list(range(0x206A, 0x2070)) + [0xFEFF] + list(range(0xFFF9, 0xFFFD)))
_C1 = {chr(b): bytes([b]).decode("cp1252") for b in range(0x80, 0xA0) if b not in (0x81, 0x8D, 0x8F, 0x90, 0x9D)}
_C1_RE = re.compile("[\x80-\x9f]")


def _fix_run(m):
    s = m.group(0)
    try:
        return bytes(_CP1252[c] for c in s).decode("utf-8")
    except (UnicodeDecodeError, KeyError):
        return s


def fix_encoding(text):
    """Undo UTF-8-read-as-cp1252 / latin-1 mojibake, run by run, up to three layers deep."""
    for _ in range(3):
        fixed = _MOJIBAKE.sub(_fix_run, text)
        if fixed == text:
            break
        text = fixed
    return text


def fix_text(text):
    """The default `ftfy.fix_text(text)` pipeline as described in the module docstring."""
    if text.isascii() and "\x1b" not in text and "\r" not in text:
        return text
    for _ in range(4):
        start = text
        text = fix_encoding(text)
        text = _C1_RE.sub(lambda m: _C1.get(m.group(0), m.group(0)), text)
        text = text.translate(_TRANSLATE)
        text = _LINE_BREAKS.sub("\n", text)
        text = _ANSI.sub("", text)
        text = text.translate(_CONTROL)
        text = unicodedata.normalize("NFC", text)
        if text == start:
            break
    return text
