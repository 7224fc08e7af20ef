"""Per-tensor gradient errors of TrainStep.backward against the reference-autograd fixture (debug print, no asserts)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import summarize, synth_sd, GOLDEN
from msclip_amd import synth, train
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
name = "b32-yfcc-msclips"
BN = os.environ.get("BN", "frozen")
g = np.load(os.path.join(GOLDEN, name + (".grads_trainbn.npz" if BN == "batch" else ".grads.npz")))
m = get_clip_model(named_config(name)); m.load_state_dict(synth_sd(name), strict=True); m = m.cuda().eval()
ts = train.TrainStep(m, lr=1e-4, bn=BN)
b = int(g["batch"])
img = synth.synth_images(b, seed=int(g["seed"])).cuda(); tok = synth.synth_tokens(b, seed=int(g["seed"]) + 1).cuda()
print("loss", ts.forward(img, tok).item(), float(g["loss"]))
grads = ts.backward()
pat = sys.argv[1:] or ["conv", "bn", "downsample", "top2bottom"]
for k in [k[2:] for k in g.files if k.startswith("g_")]:
    if not any(p in k for p in pat):
        continue
    if k not in grads:
        print("MISSING", k); continue
    ref = g["g_" + k]; sm = summarize(grads[k]); sc = max(float(g["gmax_" + k]), 1e-12)
    err = np.abs(sm[2:] - ref[2:]).max() / sc
    am = abs(sm[1] - ref[1]) / (ref[1] + 1e-12)
    cos = ""
    if "gfull_" + k in g.files:
        full = torch.from_numpy(g["gfull_" + k]).flatten()
        cos = "cos %.4f" % torch.nn.functional.cosine_similarity(grads[k].float().cpu().flatten(), full, dim=0).item()
    flag = "  <<<<" if err > 0.08 or am > 0.05 else ""
    print(f"{k:95s} err {err:.4f} absmean-dev {am:.4f} {cos} shape {tuple(grads[k].shape)}{flag}")
