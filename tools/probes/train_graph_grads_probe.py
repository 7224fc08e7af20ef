"""Companion of train_graph_probe.py: forward + backward captured by torch.cuda.graph, every gradient of the replay against the eager
backward from the same state (full caption rows).  Run with MSCLIP_COLSUM_MAIN=1 or MSCLIP_IM2COL_MAIN=1 to see the lane-stream
interaction described in profiles/r06_train_hipgraph_probe.txt disappear."""
import os, sys, time
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import load_schema
from msclip_amd import synth, train
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
name, B = "b32-yfcc-msclips", 512
img, tok = synth.synth_images(B, seed=10).cuda(), synth.synth_tokens(B, seed=100).cuda()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream()); torch.cuda.set_stream(s)
m = get_clip_model(named_config(name)); m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True); m = m.cuda().eval()
m.engine().opt = m.engine().opt.replace(text_pack=False)
ts = train.TrainStep(m, lr=1e-5, bn="frozen")
for _ in range(2):
    l = ts.forward(img, tok); ts.step(ts.backward())
l = ts.forward(img, tok); ge = {k: v.float().clone() for k, v in ts.backward().items()}
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    lg = ts.forward(img, tok)
    gg = ts.backward()
for rep in range(2):
    g.replay(); torch.cuda.synchronize()
    bad = [k for k, v in gg.items() if not torch.isfinite(v).all()]
    diff = sorted(((ge[k] - gg[k].float()).abs().max().item() / (ge[k].abs().max().item() + 1e-20), k) for k in ge if k not in bad)
    print("replay", rep, "loss", float(lg), "eager", float(l), "non-finite grads:", len(bad), bad[:12])
    print("   worst finite deviations:", diff[-4:], "bitwise equal:", sum(1 for d, _ in diff if d == 0), "of", len(ge))
