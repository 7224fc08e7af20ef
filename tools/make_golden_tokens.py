"""Token-id fixtures from the REAL reference tokenizer (build container only; needs /root/reference and its vocab).
Writes tests/golden/tokenizer.json = {"prompts": [...], "ids": [[...77 ints...], ...]}."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import as R
R.import_reference_module()                      # installs the ftfy stub and the lib/ path
from dataset.languages.simple_tokenizer import SimpleTokenizer
tok = SimpleTokenizer()
classes = ["tench", "goldfish", "great white shark", "toilet paper", "Yorkshire terrier", "jack-o'-lantern", "CD player"]
templates = ["a photo of a {}.", "a bad photo of a {}.", "itap of the {}.", "a {} in a video game.", "art of the {}."]
prompts = [t.format(c) for c in classes[:3] for t in templates[:3]] + [
    "A photo of   many  words, with punctuation!!! and numbers 12345 & symbols #@$", "it's the dog's ball; they're here, we've won",
    "&lt;html&gt; &amp;amp; entities", "naïve café façade ünïcödé", "", "x", " ".join(["word"] * 100),
    "a photo of a " + classes[5] + ".", "a photo of a " + classes[6] + ".", "UPPER lower MiXeD",
    # well-formed NFC text that ftfy.fix_text leaves unchanged (the reference's cleaner, simple_tokenizer.py:54-57; the
    # build container has no ftfy, its stub is the identity): byte-level BPE of multi-byte UTF-8
    "ein Foto von einem Bären im Schnee, ganz nah", "una foto de un niño pequeño en la playa", "фото собаки на пляже",
    "Ελληνικά γράμματα σε μια πινακίδα", "犬の写真、公園で", "صورة قطة صغيرة", "crème brûlée & piña colada",
    "a photo of a dog 😀 smiling", "Smørrebrød på Åre ÆØÅ"]
ids = tok(prompts).tolist()
json.dump({"prompts": prompts, "ids": ids, "sot": tok.get_sot_token(), "eot": tok.get_eot_token()},
          open(os.path.join(ROOT, "tests", "golden", "tokenizer.json"), "w"))
print(len(prompts), ids[0][:12])
