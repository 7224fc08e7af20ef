"""Eager vs hipGraph replay at the zero-shot loop's batch sizes (GPU box only)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msclip_amd import synth
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
name = "b32-yfcc-msclips"
schema = [(k, tuple(s), getattr(torch, d)) for k, s, d in json.load(open(f"tests/golden/{name}.schema.json"))]
m = get_clip_model(named_config(name)); m.load_state_dict(synth.synth_state_dict(schema)); m = m.cuda().eval()
eng = m.engine()
for (Bi, Bt) in [(32, 0), (0, 80), (8, 8)]:
    img = synth.synth_images(Bi).cuda() if Bi else None
    tok = synth.synth_tokens(Bt).cuda() if Bt else None
    rep = eng.graph(Bi, Bt)
    for fn, label in ((lambda: eng.run(img, tok), "eager"), (lambda: rep(img, tok), "graph")):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"Bi={Bi:3d} Bt={Bt:3d} {label}: {dt*1e3:7.3f} ms/call")
