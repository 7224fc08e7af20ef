"""Training step phases on the GPU clock (events on the compute stream, steady state): forward / backward / optimizer step.
    python tools/probes/phase_times.py [--bn batch|frozen] [--steps 12]"""
import argparse, os, sys
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("--bn", default="batch")
ap.add_argument("--steps", type=int, default=12)
a = ap.parse_args()
from bench import load_schema
from msclip_amd import hip, synth, train
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
name = "b32-yfcc-msclips"
m = get_clip_model(named_config(name))
m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True)
m = m.cuda().eval()
eng = m.engine()
img, tok = synth.synth_images(512, seed=10).cuda(), synth.synth_tokens(512, seed=100).cuda()
ts = train.from_config(m, named_config(name), bn=a.bn)
hip.use_compute_stream(torch.device("cuda:0")) if hasattr(hip, "use_compute_stream") else None
cap = eng.stage_captions(tok)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream()); return e
marks = []
for i in range(a.steps + 4):
    nxt = eng.stage_captions(tok)
    e0 = ev(); ts.forward(img, cap); e1 = ev(); g = ts.backward(); e2 = ev(); ts.step(g); e3 = ev()
    cap = nxt
    if i >= 4: marks.append((e0, e1, e2, e3))
torch.cuda.synchronize()
f = sum(x[0].elapsed_time(x[1]) for x in marks) / len(marks)
b = sum(x[1].elapsed_time(x[2]) for x in marks) / len(marks)
s = sum(x[2].elapsed_time(x[3]) for x in marks) / len(marks)
tot = marks[0][0].elapsed_time(marks[-1][3]) / len(marks)
print(f"bn={a.bn}: forward {f:.2f} ms, backward {b:.2f} ms, optimizer + re-pack {s:.2f} ms; step to step {tot:.2f} ms")
