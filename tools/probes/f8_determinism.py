import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import synth_sd
from msclip_amd import synth, hip
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
name = sys.argv[1] if len(sys.argv) > 1 else "b32-yfcc-msclips"
def mk(p):
    m = get_clip_model(named_config(name, ["MODEL.SPEC.PRECISION", p])); m.load_state_dict(synth_sd(name)); return m.cuda().eval()
bf, f8 = mk("bf16"), mk("fp8")
cos = torch.nn.functional.cosine_similarity
for B in (6, 6, 64, 6):
    img, tok = synth.synth_images(B, seed=33).cuda(), synth.synth_tokens(B, seed=34).cuda()
    a, b = f8.encode_text(tok), f8.encode_text(tok)
    c, d = f8.encode_image(img), f8.encode_image(img)
    rb = bf.encode_text(tok)
    ri = bf.encode_image(img)
    print(B, "repeat-equal text", torch.equal(a, b), "image", torch.equal(c, d), "cos text", cos(a, rb, dim=-1).min().item(), "image", cos(c, ri, dim=-1).min().item(),
          "per-sample text cos", [round(x, 4) for x in cos(a, rb, dim=-1)[:6].tolist()])
# joint run (both towers through the shared GEMMs) vs separate
w = f8.engine().run(img, tok); ft_joint = w["ft"].clone(); fi_joint = w["fv"].clone()
print("joint vs separate: text", (ft_joint - f8.encode_text(tok)).abs().max().item(), "image", (fi_joint - f8.encode_image(img)).abs().max().item())
