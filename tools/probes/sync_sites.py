"""Which Python lines of a training step issue synchronising HIP calls (blocking copies, .item(), ...):
torch.cuda.set_sync_debug_mode('warn') + a stack per warning, aggregated by the innermost msclip_amd frame."""
import collections
import sys
import traceback
import warnings

import torch

sys.path.insert(0, ".")
from msclip_amd import synth, train                                       # noqa: E402
from msclip_amd.config import named_config                                # noqa: E402
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model               # noqa: E402

bn = sys.argv[1] if len(sys.argv) > 1 else "frozen"
m = get_clip_model(named_config("b32-yfcc-msclips"))
m.load_state_dict(synth.synth_state_dict(synth.schema_of(m)), strict=True)
m = m.cuda().eval()
img, tok = synth.synth_images(32, seed=1).cuda(), synth.synth_tokens(32, seed=2).cuda()
ts = train.TrainStep(m, lr=1e-4, bn=bn)
for _ in range(3):
    ts.forward(img, tok)
    ts.step(ts.backward())
torch.cuda.synchronize()
sites = collections.Counter()


def show(message, category, filename, lineno, file=None, line=None):
    frames = [f for f in traceback.extract_stack() if "/msclip_amd/" in f.filename or "bench.py" in f.filename]
    key = " <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno} {f.line.strip()[:70]}" for f in reversed(frames[-2:])) if frames else "?"
    sites[(str(message)[:60], key)] += 1


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
n = 2
for _ in range(n):
    ts.forward(img, tok)
    ts.step(ts.backward())
torch.cuda.set_sync_debug_mode("default")
print(bn, "synchronising calls per step:", sum(sites.values()) / n)
for (msg, key), c in sites.most_common(40):
    print(f"{c / n:6.1f}  {msg} | {key}")
