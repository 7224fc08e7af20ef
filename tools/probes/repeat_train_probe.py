"""Bitwise repeatability of the training step's gradients across fresh, NaN-poisoned workspaces (GPU box only).
    python tools/probes/repeat_train_probe.py <model> <batch> [reps] [bn]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_sd
from msclip_amd import synth, train
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config

name, B = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
bn = sys.argv[4] if len(sys.argv) > 4 else "frozen"
m = get_clip_model(named_config(name))
m.load_state_dict(synth_sd(name), strict=True)
m = m.cuda().eval()
ts = train.TrainStep(m, lr=1e-5, bn=bn)
img, tok = synth.synth_images(B, seed=51).cuda(), synth.synth_tokens(B, seed=52).cuda()
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
ref = None
for rep in range(reps):
    m.load_state_dict(sd0, strict=True)                      # (train-mode BN updates running statistics)
    ts.eng.refresh(force=True)
    ts.eng._ws = {k: v for k, v in ts.eng._ws.items() if k == "loss_ws"}
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = torch.full((int(20e9) // 4,), float("nan"), device="cuda")
    del junk
    loss = float(ts.forward(img, tok))
    g = ts.backward()
    torch.cuda.synchronize()
    g = {k: v.float().clone() for k, v in g.items()}
    nan = [k for k, v in g.items() if not torch.isfinite(v).all()]
    if ref is None:
        ref = (loss, g)
    bad = [k for k in g if k != "token_embedding.weight" and not torch.equal(g[k], ref[1][k])]
    worst = max(((g[k] - ref[1][k]).abs().max().item() / (ref[1][k].abs().max().item() + 1e-12), k) for k in g) if bad else (0, "")
    print(f"rep {rep}: loss {loss:.6f} same {loss == ref[0]}  nan {nan[:3]}  differing gradients {len(bad)} {bad[:4]}  worst rel {worst[0]:.2e} {worst[1][-60:]}", flush=True)
