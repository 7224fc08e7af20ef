"""Host synchronisations inside one training step: torch's sync debug mode (warns at every synchronising torch call) plus
the host-side return time of refresh / forward / backward / step with a busy GPU."""
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from msclip_amd import synth, train                                  # noqa: E402
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model         # noqa: E402
from msclip_amd.config import named_config                          # noqa: E402


def load_schema(name):
    with open(os.path.join(ROOT, "tests", "golden", name + ".schema.json")) as f:
        return [(k, tuple(s), getattr(torch, d)) for k, s, d in json.load(f)]


name = "b32-yfcc-msclips"
m = get_clip_model(named_config(name))
m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True)
m = m.cuda().eval()
B = int(os.environ.get("PROBE_BATCH", "512"))
img, tok = synth.synth_images(B, seed=1).cuda(), synth.synth_tokens(B, seed=2).cuda()
ts = train.from_config(m, named_config(name), bn=sys.argv[1] if len(sys.argv) > 1 else "batch")
for _ in range(3):
    ts.forward(img, tok); ts.step(ts.backward())
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as wl:
    warnings.simplefilter("always")
    ts.forward(img, tok)
    g = ts.backward()
    ts.step(g)
    ts.forward(img, tok)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
seen = {}
for w in wl:
    key = (w.filename.replace(ROOT + "/", ""), w.lineno, str(w.message)[:90])
    seen[key] = seen.get(key, 0) + 1
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, k)
print("sync warnings:", len(wl))
ts.step(ts.backward())
torch.cuda.synchronize()
# host return times with the GPU busy (a step queued in front)
for label, fn in (("refresh", lambda: ts.eng.refresh(force=True)), ("forward", lambda: ts.forward(img, tok))):
    ts.forward(img, tok); ts.step(ts.backward())          # ~100 ms of GPU work queued
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label}: host returned after {1e3 * (t1 - t0):.1f} ms (GPU drained after {1e3 * (time.perf_counter() - t0):.1f} ms)")
    if label == "forward":
        ts.step(ts.backward())
        torch.cuda.synchronize()
