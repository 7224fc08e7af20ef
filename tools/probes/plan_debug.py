import sys
import torch
sys.path.insert(0, ".")
from msclip_amd import synth, train
from msclip_amd.config import named_config
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
m = get_clip_model(named_config("b32-yfcc-msclips"))
m.load_state_dict(synth.synth_state_dict(synth.schema_of(m)), strict=True)
m = m.cuda().eval()
ts = train.TrainStep(m, lr=3e-5, bn="batch")
img, tok = synth.synth_images(6, seed=61).cuda(), synth.synth_tokens(6, seed=62).cuda()
for i in range(3):
    ts.forward(img, tok)
    g = ts.backward()
    plan = getattr(ts, "_plan", None)
    if plan is not None:
        sig = (id(ts.eng.tblk[0]["w"].wqkv), not ts.eng.fp8, id(ts.state))
        print("step", i, "sig same", plan.sig == sig, plan.sig, sig, "ngrads", len(g), plan.ngrads)
        bad = [k for k in plan.names if not (k in g and g[k].is_contiguous() and g[k].dtype == torch.float32)]
        print("  bad", bad[:10], [(g[k].shape, g[k].stride(), g[k].dtype) for k in bad[:5] if k in g])
    ts.step(g)
