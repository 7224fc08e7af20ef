"""Time of the conv side's table-driven re-pack (msclip_pack_weights) against the tensor-algebra re-pack (GPU box only)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import synth_sd
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
name = "b32-yfcc-msclips"
m = get_clip_model(named_config(name)); m.load_state_dict(synth_sd(name)); m = m.cuda().eval()
eng = m.engine(); eng.refresh()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3, (time.perf_counter() - t0) / n * 1e6
print("msclip_pack_weights: %.1f us GPU, %.1f us wall per call (%d items, %d workgroups)" % (*t(eng._pack_plan.run), eng._pack_plan.n_items, eng._pack_plan.n_blocks))
with torch.no_grad():
    print("tensor-algebra conv-side pack: %.1f us GPU, %.1f us wall per call" % t(lambda: eng._pack(m, blocks=False), 5))
