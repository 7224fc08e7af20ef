"""hipGraph replay (Engine.graph: a capture of the launch-table replay, static input buffers) against the launch-table replay itself for
the C2 towers (b32, batch 512): 10.32-10.34 vs 10.22 ms -- the graph pays its 308 MB input copy and gains nothing on kernel-to-kernel
gaps; the table is the shipped path, the graph remains for callers who want one object to launch."""
import os, sys, time, json
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import load_schema
from msclip_amd import synth
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
name = "b32-yfcc-msclips"
m = get_clip_model(named_config(name)); m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True); m = m.cuda().eval()
eng = m.engine()
B = 512
img, tok = synth.synth_images(B, seed=10).cuda(), synth.synth_tokens(B, seed=100).cuda()
replay = eng.graph(B, B)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / n * 1e3, 3)
res = {"plan": [], "graph": []}
for r in range(3):
    res["plan"].append(t(lambda: eng.run(img, tok)))
    res["graph"].append(t(lambda: replay(img, tok)))
print(json.dumps(res))
