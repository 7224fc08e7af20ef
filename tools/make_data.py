"""Extract the DATA files the text side needs from the reference tree (build container only):

  msclip_amd/data/clip_bpe_merges.txt.gz   the 48 894 byte-pair merges of the 49 408-entry CLIP vocabulary (the rows of the
                                           reference's lib/dataset/languages/bpe_simple_vocab_16e6.txt.gz that its tokenizer
                                           actually reads, simple_tokenizer.py:71 -- the released checkpoints' token ids are
                                           defined by this table, it cannot be regenerated)
  msclip_amd/data/imagenet_prompts.json    the 1000 ImageNet class names and 80 prompt templates of the zero-shot protocol
                                           (values of lib/dataset/prompts/constants.py)

Both are data (tables of strings), not code; nothing executable of the reference is copied.

    python tools/make_data.py
"""
import gzip
import json
import os
import runpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MSCLIP_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "msclip_amd", "data")


def main():
    os.makedirs(OUT, exist_ok=True)
    with gzip.open(os.path.join(REF, "lib", "dataset", "languages", "bpe_simple_vocab_16e6.txt.gz"), "rt", encoding="utf-8") as f:
        lines = f.read().split("\n")
    n = 49408 - 256 - 256 - 2
    merges = lines[1:1 + n]
    assert len(merges) == n and all(len(m.split()) == 2 for m in merges)
    with gzip.GzipFile(os.path.join(OUT, "clip_bpe_merges.txt.gz"), "wb", mtime=0) as f:
        f.write(("#msclip_amd merges v1: %d pairs\n" % n + "\n".join(merges) + "\n").encode("utf-8"))
    ns = runpy.run_path(os.path.join(REF, "lib", "dataset", "prompts", "constants.py"))
    classes = list(ns["ALL_CLASSES_DICT"]["imagenet"]) if "ALL_CLASSES_DICT" in ns else list(ns["IMAGENET_CLASSES"])
    templates = ns["ALL_TEMPLATES_DICT"]["imagenet"] if "ALL_TEMPLATES_DICT" in ns else ns["IMAGENET_DEFAULT_TEMPLATES"]
    templates = [t if isinstance(t, str) else t("{}") for t in templates]
    assert len(classes) == 1000 and len(templates) == 80, (len(classes), len(templates))
    with open(os.path.join(OUT, "imagenet_prompts.json"), "w") as f:
        json.dump({"dataset": "imagenet", "classes": classes, "templates": templates}, f, indent=0)
    print("merges", n, "classes", len(classes), "templates", len(templates))


if __name__ == "__main__":
    main()
