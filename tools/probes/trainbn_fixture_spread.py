"""The train-mode BatchNorm gradient fixture's conv-side metrics under numerically EQUIVALENT paths of the step (option switches that
change summation orders / rounding points only): how far the worst-tensor metric moves between equally valid computations."""
import itertools
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from conftest import GOLDEN, summarize, synth_sd
    from msclip_amd import options, synth, train
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    import test_gpu_train as T
    for name in ("b16-yfcc-msclips", "b32-yfcc-msclips"):
        g = np.load(os.path.join(GOLDEN, name + ".grads_trainbn.npz"))
        b = int(g["batch"])
        img = synth.synth_images(b, seed=int(g["seed"])).cuda()
        tok = synth.synth_tokens(b, seed=int(g["seed"]) + 1).cuda()
        expect = [k[2:] for k in g.files if k.startswith("g_")]
        conv_keys = [k for k in expect if any(f in k for f in T.CONV_SIDE)]
        for views, two, fused in itertools.product((False, True), repeat=3):
            options.TRAIN = options.TrainOptions.from_env().replace(adapter_bn_views=views, bn_two_pass=two, bn_bwd_fused=fused)
            m = get_clip_model(named_config(name))
            m.load_state_dict(synth_sd(name), strict=True)
            m = m.cuda().eval()
            ts = train.TrainStep(m, lr=1e-4, bn="batch")
            ts.forward(img, tok)
            grads = ts.backward()
            worst = {k: float(np.abs(summarize(grads[k])[2:] - g["g_" + k][2:]).max() / max(float(g["gmax_" + k]), 1e-12)) for k in expect}
            cos = min(F.cosine_similarity(grads[k].float().cpu().flatten(), torch.from_numpy(g["gfull_" + k]).flatten(), dim=0).item()
                      for k in conv_keys if "gfull_" + k in g.files)
            wk = max(conv_keys, key=lambda k: worst[k])
            print(f"{name} batch {b} views={int(views)} two_pass={int(two)} fused_bwd={int(fused)}: conv-side worst {worst[wk]:.4f} ({wk[-60:]}) "
                  f"median {np.median([worst[k] for k in conv_keys]):.4f} lowest cosine {cos:.4f}", flush=True)


if __name__ == "__main__":
    main()
