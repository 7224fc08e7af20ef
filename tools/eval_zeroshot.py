"""Zero-shot evaluation CLI, counterpart of the reference's tools/eval_zeroshot.py + tools/zero_shot.py.

    python tools/eval_zeroshot.py --model experiments/model/b32-yfcc-msclips.yaml --val-root DATASET/imagenet/val \
        --prompts lib/dataset/prompts/constants.py [--ckpt OUTPUT_MODEL/b32-yfcc-msclips_ckpt.pth] [--bpe vocab.gz] \
        [--max-images 64] [opts KEY VALUE ...]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msclip_amd import checkpoint, zeroshot                       # noqa: E402
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model      # noqa: E402
from msclip_amd.config import load_config                         # noqa: E402
from msclip_amd.tokenizer import SimpleTokenizer                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True, help="experiments/model/*.yaml")
    ap.add_argument("--val-root", required=True, help="ImageFolder root (val/<wnid>/*.JPEG)")
    ap.add_argument("--prompts", required=True, help="json or python file with class names and prompt templates")
    ap.add_argument("--ckpt", default=None, help="defaults to MODEL.PRETRAINED_MODEL of the yaml")
    ap.add_argument("--bpe", default=None)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--max-images", type=int, default=None)
    ap.add_argument("opts", nargs=argparse.REMAINDER)
    args = ap.parse_args()

    cfg = load_config(args.model, args.opts or None)
    model = get_clip_model(cfg)
    ckpt = args.ckpt or cfg.MODEL.PRETRAINED_MODEL
    if ckpt and os.path.isfile(ckpt):
        checkpoint.load_pretrained(model, ckpt)
        print(f"=> loaded {ckpt}")
    else:
        print(f"=> WARNING: checkpoint {ckpt!r} not found, evaluating random-init weights")
    model = model.cuda().eval()
    classes, templates = zeroshot.load_prompts(args.prompts)
    tok = SimpleTokenizer(args.bpe)
    bs = args.batch_size or cfg.TEST.BATCH_SIZE_PER_GPU
    res = zeroshot.evaluate(model, tok, args.val_root, classes, templates, batch_size=bs, max_images=args.max_images,
                            size=cfg.TEST.IMAGE_SIZE[0])
    print(res)


if __name__ == "__main__":
    main()
