"""Time the zero-shot classifier build (reference tools/zero_shot.py:122-134: 1000 ImageNet classes x 80 prompt templates =
80 000 captions through encode_text) with packed and full-row captions.  Prompts are tokenized once up front (the BPE
tokenizer is host work and the same for both); what is timed is the 125 encode_text calls of 640 prompts + the per-class mean.

    python tools/zeroshot_classifier_time.py [--model b32-yfcc-msclips] [--classes 1000]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="b32-yfcc-msclips")
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--per-batch", type=int, default=8)
    a = ap.parse_args()
    from bench import load_schema
    from msclip_amd import synth, zeroshot
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    from msclip_amd.tokenizer import SimpleTokenizer
    m = get_clip_model(named_config(a.model))
    m.load_state_dict(synth.synth_state_dict(load_schema(a.model), seed=0), strict=True)
    m = m.cuda().eval()
    eng = m.engine()
    classes, templates = zeroshot.load_prompts("imagenet")
    classes = classes[:a.classes]
    tok = SimpleTokenizer()
    t0 = time.perf_counter()
    batches = []
    for c0 in range(0, len(classes), a.per_batch):
        group = classes[c0:c0 + a.per_batch]
        batches.append(tok([t.format(c) for c in group for t in templates]).cuda())
    t_tok = time.perf_counter() - t0
    lens = torch.cat([(b.argmax(-1) + 1) for b in batches]).float()
    out = {"model": a.model, "prompts": int(lens.numel()), "mean_live_rows": round(lens.mean().item(), 2), "max_live_rows": int(lens.max()),
           "tokenizer_s": round(t_tok, 2)}
    res = {}
    for mode in ("0", "1", "0", "1"):
        eng.opt = eng.opt.replace(text_pack=(mode == "1"))
        for b in batches[:3]:
            m.encode_text(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cols = []
        for b in batches:
            emb = m.encode_text(b).float().reshape(-1, len(templates), m.embed_dim).mean(dim=1)
            cols.append(emb / emb.norm(dim=-1, keepdim=True))
        W = torch.cat(cols, 0).t().contiguous()
        torch.cuda.synchronize()
        res.setdefault(mode, []).append((time.perf_counter() - t0, W))
    full, packed = min(t for t, _ in res["0"]), min(t for t, _ in res["1"])
    dW = (res["0"][0][1] - res["1"][0][1]).abs().max().item()
    out.update(full_rows_s=round(full, 3), packed_s=round(packed, 3), speedup=round(full / packed, 2), classifier_max_abs_diff=dW)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
