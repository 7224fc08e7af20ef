"""Un-profiled host time of the training step's three calls against the step time (is the host ahead of the GPU?)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from msclip_amd import hip, synth, train                                  # noqa: E402
from msclip_amd.config import named_config                                # noqa: E402
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model               # noqa: E402

bn = sys.argv[1] if len(sys.argv) > 1 else "frozen"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
m = get_clip_model(named_config("b32-yfcc-msclips"))
m.load_state_dict(synth.synth_state_dict(synth.schema_of(m)), strict=True)
m = m.cuda().eval()
img, tok = synth.synth_images(B, seed=1).cuda(), synth.synth_tokens(B, seed=2).cuda()
ts = train.from_config(m, named_config("b32-yfcc-msclips"), bn=bn)
for _ in range(4):
    ts.forward(img, tok)
    ts.step(ts.backward())
torch.cuda.synchronize()
n = 8
tf = tb = tsx = 0.0
t00 = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter()
    ts.forward(img, tok)
    t1 = time.perf_counter()
    g = ts.backward()
    t2 = time.perf_counter()
    ts.step(g)
    t3 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1; tsx += t3 - t2
t_issue = time.perf_counter() - t00
torch.cuda.synchronize()
t_all = time.perf_counter() - t00
print(f"{bn} B={B}: host forward {1e3 * tf / n:.1f} ms, backward {1e3 * tb / n:.1f} ms, step {1e3 * tsx / n:.1f} ms; "
      f"host total {1e3 * t_issue / n:.1f} ms per step, wall incl. final drain {1e3 * t_all / n:.1f} ms per step")
