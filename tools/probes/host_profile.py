"""cProfile of the training step's host side (which Python / ATen calls the step boundary spends its time in).
    python tools/probes/host_profile.py [frozen|batch] [B]"""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from msclip_amd import synth, train                                       # noqa: E402
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
# un-profiled host time of the three calls
n = 6
acc = [0.0, 0.0, 0.0]
for _ in range(n):
    t0 = time.perf_counter(); ts.forward(img, tok)
    t1 = time.perf_counter(); g = ts.backward()
    t2 = time.perf_counter(); ts.step(g)
    t3 = time.perf_counter()
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
torch.cuda.synchronize()
print(f"host ms per step: forward {1e3 * acc[0] / n:.2f}, backward {1e3 * acc[1] / n:.2f}, optimizer step {1e3 * acc[2] / n:.2f}")
for name, fn in (("step", None), ("forward", None)):
    pr = cProfile.Profile()
    for _ in range(n):
        g = None
        if name == "forward":
            pr.enable(); ts.forward(img, tok); pr.disable()
            g = ts.backward()
            ts.step(g)
        else:
            ts.forward(img, tok)
            g = ts.backward()
            pr.enable(); ts.step(g); pr.disable()
    torch.cuda.synchronize()
    print(f"==== {name}: {n} calls")
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)

# the optimizer step's host time with an EMPTY queue (is the time in AdamwPlan.run launch cost or back-pressure?)
pr = cProfile.Profile()
tt = 0.0
for _ in range(n):
    ts.forward(img, tok)
    g = ts.backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable(); ts.step(g); pr.disable()
    tt += time.perf_counter() - t0
torch.cuda.synchronize()
print(f"==== step with the GPU idle: host {1e3 * tt / n:.2f} ms per call")
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
