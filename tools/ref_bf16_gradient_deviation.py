"""How far do the REFERENCE's own gradients move when its forward / backward run under bf16 autocast?  (build container
only: imports /root/reference through tools/ref_import.py)

The training step of this build computes with bf16 GEMM operands and bf16 activations / gradient maps; its parity tests
compare against fp32 autograd of the reference with stated tolerances.  This script measures the yardstick for those
tolerances: the reference itself, same weights, same batches, once in fp32 and once under torch.autocast(bfloat16), and
the per-tensor deviation in the very metrics the tests use (error over a 64-point sample relative to the tensor's
abs-max, abs-mean deviation, cosine), grouped like the tests group them.  Output: tests/golden/ref_bf16_gradient_deviation.json
(numbers only)."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import as R                      # noqa: E402
from msclip_amd import synth                # noqa: E402
from make_golden import summarize, SEED     # noqa: E402

CONV_SIDE = ("resblocks.0.conv1", "resblocks.0.bn1", "resblocks.0.resnet_stage", "resblocks.0.last_conv", "parallel_branch_v",
             "top2bottom", "bottom_dw_conv")
LNB = ("ln_1.bias", "ln_2.bias", "ln_final.bias", "ln_post.bias", "ln_pre.bias", "ln_adapt.bias")


def grads(model, img, tok, train_bn, autocast):
    for p in model.parameters():
        p.grad = None
    before = {k: v.clone() for k, v in model.state_dict().items()}
    model.train(train_bn)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        logits = model(img, tok)
    logits = logits.float()
    lab = torch.arange(img.shape[0])
    loss = 0.5 * (F.cross_entropy(logits, lab) + F.cross_entropy(logits.t(), lab))
    loss.backward()
    out, seen = {}, set()
    for k, p in model.named_parameters(remove_duplicate=False):
        if p.grad is not None and id(p) not in seen:
            seen.add(id(p))
            out[k] = p.grad.detach().float().clone()
    model.load_state_dict(before)            # running statistics back
    model.eval()
    return float(loss), out


def compare(ref, got):
    rows = {}
    for k, r in ref.items():
        g = got[k]
        sr, sg = summarize(r), summarize(g)
        scale = max(r.abs().max().item(), 1e-12)
        rows[k] = (float(np.abs(sg[2:] - sr[2:]).max() / scale), float(abs(sg[1] - sr[1]) / (sr[1] + 1e-12)),
                   float(F.cosine_similarity(g.flatten(), r.flatten(), dim=0)))
    def cls(k):
        return "ln_bias" if k.endswith(LNB) else "conv_side" if any(f in k for f in CONV_SIDE) else "token_side"
    res = {}
    for c in ("token_side", "conv_side", "ln_bias"):
        v = [rows[k] for k in rows if cls(k) == c]
        res[c] = {"tensors": len(v), "sample_err_median": float(np.median([x[0] for x in v])),
                  "sample_err_worst": float(max(x[0] for x in v)), "absmean_dev_worst": float(max(x[1] for x in v)),
                  "cosine_lowest": float(min(x[2] for x in v))}
    return res


def run_model(name, cases):
    model, _ = R.build_reference_model(name)
    model.load_state_dict(synth.synth_state_dict(synth.schema_of(model), seed=SEED), strict=True)
    for p in model.parameters():
        p.requires_grad_(True)
    out = {}
    for tag, batch, train_bn in cases:
        img, tok = synth.synth_images(batch, seed=SEED), synth.synth_tokens(batch, seed=SEED + 1)
        l32, g32 = grads(model, img, tok, train_bn, False)
        l16, g16 = grads(model, img, tok, train_bn, True)
        out[tag] = dict(loss_fp32=l32, loss_bf16=l16, **compare(g32, g16))
        print(name, tag, json.dumps(out[tag], indent=1), flush=True)
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(16)
    R.ensure_single_rank_group()
    path = os.path.join(ROOT, "tests", "golden", "ref_bf16_gradient_deviation.json")
    only = sys.argv[1] if len(sys.argv) > 1 else None
    out = json.load(open(path)) if only and os.path.exists(path) else {}
    out["what"] = ("reference gradients under torch.autocast(bfloat16) against the same reference in fp32 (CPU), metrics of "
                   "tests/test_gpu_train.py; top-level tags: b32-yfcc-msclips, other models under their name; the batches "
                   "are the gradient fixtures' (tools/make_golden.py::grads_fixture)")
    out["model"] = "b32-yfcc-msclips"
    if only == "b32-batch32":               # round 6: the yardstick of the batch-32 gradient fixtures (tools/make_golden.py --grads-b32)
        out.update(run_model("b32-yfcc-msclips", (("eval_bn_batch32", 32, False), ("train_bn_batch32", 32, True))))
    if only in (None, "b32-yfcc-msclips"):
        out.update(run_model("b32-yfcc-msclips", (("eval_bn_batch4", 4, False), ("train_bn_batch16", 16, True))))
    if only in (None, "b16-yfcc-msclips"):
        out["b16-yfcc-msclips"] = run_model("b16-yfcc-msclips", (("eval_bn_batch4", 4, False), ("train_bn_batch8", 8, True)))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
