"""north_star's accuracy acceptance ("zero-shot IN-1K top-1 within +-0.1 of the reference checkpoint", reference README.md:26-28,
tools/zero_shot.py:265-275) cannot be measured offline: neither the released checkpoint nor ImageNet exists here.  What CAN be
measured is the only way the HIP bf16 path could move top-1: by flipping the arg-max of `100 * f_img @ W` on images whose two best
classes are nearly tied.  This test measures that flip rate at full scale -- the packaged 1000 classes x 80 templates through the
real tokenizer, 4096 generated images -- against the fp32 oracle (the pinned restatement of the reference, run on the GPU in fp32
as the CHECKER), and checks the mechanism: every flip sits inside a margin band of twice the largest logit deviation.

One-command run on the real checkpoint (when a user has it):
    python tools/eval_zeroshot.py --ds imagenet --model experiments/model/b32-yfcc-msclips.yaml --ckpt b32_yfcc_msclips_ckpt.pth \
        DATASET.ROOT /path/to/imagenet
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, synth_sd
from msclip_amd import synth, zeroshot
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
from msclip_amd.tokenizer import SimpleTokenizer
from oracle import msclip_oracle as O

pytestmark = pytest.mark.gpu

N_IMAGES = 4096            # HIP path
N_ORACLE = 1024            # fp32 oracle subsample (>= 512)
FLIP_BOUND = 0.08          # measured 0.053 on the synthetic weights, whose 1000 class logits are nearly tied (std 0.83 logit units over
#                            the classes, median top-1 / top-2 margin 0.07): profiles/r06_accuracy_flip_proxy.json
FLIP_BAND = 0.05           # every flip sits on an image whose two best oracle logits are closer than this (measured: 0.024)


def test_argmax_flip_rate_against_the_fp32_oracle(gpu_device):
    name = "b32-yfcc-msclips"
    sd = synth_sd(name)
    m = get_clip_model(named_config(name))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    sd_gpu = {k: v.cuda() for k, v in sd.items()}
    arch = O.arch_b32()
    classes, templates = zeroshot.load_prompts("imagenet")
    assert len(classes) == 1000 and len(templates) == 80
    tok = SimpleTokenizer()
    with torch.no_grad():
        W_hip = zeroshot.zeroshot_classifier(m, tok, classes, templates, classes_per_batch=16).float()        # [512, 1000]
        cols = []
        for c in classes:                                                                                     # oracle: 80 000 prompts, fp32
            e = O.encode_text(tok([t.format(c) for t in templates]).cuda(), sd_gpu, arch).mean(dim=0)
            cols.append(e / e.norm())
        W_ref = torch.stack(cols, dim=1)
        f_hip = torch.cat([m.encode_image(synth.synth_images(256, seed=900 + i).cuda()).float() for i in range(N_IMAGES // 256)])
        f_ref = torch.cat([O.encode_image(synth.synth_images(256, seed=900 + i).cuda(), sd_gpu, arch) for i in range(N_ORACLE // 256)])
    lg_hip = 100.0 * f_hip @ W_hip
    lg_ref = 100.0 * f_ref @ W_ref
    dev = (lg_hip[:N_ORACLE] - lg_ref).abs()
    eps = dev.max().item()
    top2 = lg_ref.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    flips = lg_hip[:N_ORACLE].argmax(1) != lg_ref.argmax(1)
    flip_rate = flips.float().mean().item()
    band = (margin <= 2 * eps).float().mean().item()
    # the same question inside the HIP path at 4096 images: its own classifier vs the oracle's classifier (isolates the text side)
    flips_w = (lg_hip.argmax(1) != (100.0 * f_hip @ W_ref).argmax(1)).float().mean().item()
    hist_edges = [0.0, 0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0, 2.0, 5.0, 1e9]
    hist = torch.histogram(margin.cpu(), torch.tensor(hist_edges)).hist.tolist()
    rec = {"model": name, "weights": "synthetic (deterministic random init)", "classes": 1000, "templates": 80, "images_hip": N_IMAGES,
           "images_oracle": N_ORACLE, "logit_scale": 100.0, "max_abs_logit_deviation": eps, "mean_abs_logit_deviation": dev.mean().item(),
           "classifier_max_abs_deviation": (W_hip - W_ref).abs().max().item(),
           "argmax_flip_rate_vs_fp32_oracle": flip_rate, "flips": int(flips.sum()),
           "fraction_of_images_with_margin_below_2eps": band, "flip_rate_from_classifier_alone_4096": flips_w,
           "largest_margin_of_a_flipped_image": margin[flips].max().item() if flips.any() else 0.0,
           "fraction_of_images_inside_the_flip_band": (margin <= FLIP_BAND).float().mean().item(),
           "reading": ("a decision flips only where the oracle's two best classes are closer than ~0.025 logit units (about the mean "
                       "logit deviation); the top-1 difference to the reference on a data set is therefore at most the fraction of its "
                       "images with such a margin (here 16 % because random-init logits are nearly tied; a trained CLIP spreads its "
                       "class logits ~10 x wider), and flips move top-1 in both directions"),
           "oracle_margin_histogram": {"edges": hist_edges[:-1] + ["inf"], "counts": hist},
           "median_margin": margin.median().item(), "logit_std_over_classes": lg_ref.std(dim=1).mean().item()}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "accuracy_flip_proxy.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))
    assert eps <= 0.3                                                    # the stated x100 zero-shot logit tolerance
    assert (W_hip - W_ref).abs().max().item() <= 5e-3
    # mechanism: a flip needs the two best oracle logits to be closer than the two deviations can bridge
    assert not flips.any() or margin[flips].max().item() <= 2 * eps + 1e-6
    assert flip_rate <= band + 1e-9
    assert not flips.any() or margin[flips].max().item() <= FLIP_BAND, rec
    assert flip_rate <= FLIP_BOUND, rec
