"""CPU-side cost of the optimizer prelude and of Engine._pack after a training step (host time, nothing synchronised in between)."""
import collections
import sys
import time

import torch

sys.path.insert(0, ".")
from msclip_amd import engine as E, hip, packing as P, synth, train         # noqa: E402
from msclip_amd.config import named_config                                # noqa: E402
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model               # noqa: E402

m = get_clip_model(named_config("b32-yfcc-msclips"))
m.load_state_dict(synth.synth_state_dict(synth.schema_of(m)), strict=True)
m = m.cuda().eval()
img, tok = synth.synth_images(64, seed=1).cuda(), synth.synth_tokens(64, seed=2).cuda()
ts = train.TrainStep(m, lr=1e-4, bn=sys.argv[1] if len(sys.argv) > 1 else "frozen")
for _ in range(3):
    ts.forward(img, tok)
    ts.step(ts.backward())
torch.cuda.synchronize()
acc = collections.defaultdict(float)


def timed(mod, name):
    fn = getattr(mod, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[f"{mod.__name__.split('.')[-1]}.{name}"] += time.perf_counter() - t0
    setattr(mod, name, w)


for name in ("bn_fold_all", "bottleneck", "stem_stage", "adapter_weights", "stem_dual_weights", "qkv_weights"):
    timed(P, name)
timed(E, "_BlockW")
timed(E, "_LN")
timed(hip, "adamw_multi")
n = 5
tot = collections.defaultdict(float)
for _ in range(n):
    ts.forward(img, tok)
    g = ts.backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pg = ts.param_groups()
    t1 = time.perf_counter()
    eng_refresh = ts.eng.refresh
    ts.eng.refresh = lambda force=False: None
    rao = ts.eng.repack_after_optimizer
    ts.eng.repack_after_optimizer = lambda: None
    ts.step(g)
    t2 = time.perf_counter()
    ts.eng.refresh = eng_refresh
    ts.eng.repack_after_optimizer = rao
    sd = {k: t.detach() for k, t in m.state_dict().items()}
    t3 = time.perf_counter()
    ts.eng.repack_after_optimizer()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    tot["param_groups()"] += t1 - t0; tot["step() without refresh"] += t2 - t1; tot["state_dict()"] += t3 - t2
    tot["repack_after_optimizer host"] += t4 - t3; tot["... until the GPU is idle"] += t5 - t4
for k, v in tot.items():
    print(f"{k:32s} {v / n * 1e3:7.2f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"   {k:29s} {v / n * 1e3:7.2f} ms / step")
