"""Where do the ATen launches of one training step come from?  (probe: torch.profiler with Python stacks)"""
import collections, os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import synth_sd
from msclip_amd import synth, train
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
name, bn = "b32-yfcc-msclips", sys.argv[1] if len(sys.argv) > 1 else "frozen"
m = get_clip_model(named_config(name)); m.load_state_dict(synth_sd(name)); m = m.cuda().eval()
ts = train.from_config(m, named_config(name), bn=bn)
img, tok = synth.synth_images(64, seed=1).cuda(), synth.synth_tokens(64, seed=2).cuda()
for _ in range(2):
    ts.forward(img, tok); ts.step(ts.backward())
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    ts.forward(img, tok); ts.step(ts.backward())
    torch.cuda.synchronize()
sites = collections.Counter()
LAUNCH = {"aten::mul", "aten::mul_", "aten::add", "aten::add_", "aten::sub", "aten::copy_", "aten::fill_", "aten::zero_", "aten::sum", "aten::rsqrt",
          "aten::clone", "aten::contiguous", "aten::cat", "aten::div", "aten::div_", "aten::clamp_min_", "aten::mm", "aten::matmul", "aten::outer",
          "aten::_to_copy", "aten::exp", "aten::sqrt", "aten::neg", "aten::repeat", "aten::amax", "aten::abs", "aten::_foreach_copy_", "aten::lerp_"}
shown = 0
for ev in prof.events():
    if ev.name not in LAUNCH or (ev.cpu_parent is not None and ev.cpu_parent.name in LAUNCH):
        continue
    if shown < 2:
        print("sample stack:", ev.name, (ev.stack or [])[:8]); shown += 1
    st = [s for s in (ev.stack or []) if "msclip_amd" in s or "bench.py" in s]
    sites[(st[0].split("msclip_amd/")[-1][:70] if st else "?", ev.name)] += 1
tot = sum(sites.values())
print("top-level aten ops per step:", tot)
agg = collections.Counter()
for (site, op), n in sites.items():
    agg[site] += n
for site, n in agg.most_common(45):
    ops = ", ".join(f"{op.replace('aten::','')}x{c}" for (s2, op), c in sorted(sites.items(), key=lambda kv: -kv[1]) if s2 == site)[:110]
    print(f"{n:5d}  {site:60s} {ops}")
