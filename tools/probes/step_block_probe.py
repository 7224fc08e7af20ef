"""Where does the host block at the training step's boundary?  Times the pieces of TrainStep.step in the steady-state loop."""
import sys
import time

import torch

sys.path.insert(0, ".")
from msclip_amd import hip, synth, train                                  # noqa: E402
from msclip_amd.config import named_config                                # noqa: E402
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model               # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
m = get_clip_model(named_config("b32-yfcc-msclips"))
m.load_state_dict(synth.synth_state_dict(synth.schema_of(m)), strict=True)
m = m.cuda().eval()
B = 512
img, tok = synth.synth_images(B, seed=1).cuda(), synth.synth_tokens(B, seed=2).cuda()
ts = train.from_config(m, named_config("b32-yfcc-msclips"), bn="frozen")
T = {}
x = torch.randn(256, 768, device="cuda"); g_ = torch.ones(768, device="cuda"); b_ = torch.zeros(768, device="cuda")
lo = torch.empty(256, 768, dtype=torch.bfloat16, device="cuda")


def wrap(obj, name, key):
    f = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        T[key] = T.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)


orig_run = hip.AdamwPlan.run


def run(self, *a):
    if mode == "dummy":                             # a small launch first: does the block move to it?
        t0 = time.perf_counter(); hip.layernorm(x, g_, b_, lo, 256); T["dummy launch before run"] = T.get("dummy launch before run", 0.0) + time.perf_counter() - t0
    if mode == "query":
        t0 = time.perf_counter(); torch.cuda.current_stream().query(); T["stream.query before run"] = T.get("stream.query before run", 0.0) + time.perf_counter() - t0
    if mode == "fine":
        L = hip.lib()
        st = hip._stream()
        t0 = time.perf_counter()
        with torch.cuda.device(self.device):
            t1 = time.perf_counter()
            rc = L.msclip_adamw_multi(self.arr, self.n, *a, st)
            t2 = time.perf_counter()
        t3 = time.perf_counter()
        assert rc == 0
        for k, v in (("run: ctx enter", t1 - t0), ("run: msclip_adamw_multi", t2 - t1), ("run: ctx exit", t3 - t2)):
            T[k] = T.get(k, 0.0) + v
        return
    if mode == "split":                              # the same table in pieces of 4 tensors: many small calls
        import ctypes
        L = hip.lib()
        st = hip._stream()
        t0 = time.perf_counter()
        worst = 0.0
        for i in range(0, self.n, 4):
            sub = (hip.AdamwTensor * 4).from_address(ctypes.addressof(self.arr) + i * ctypes.sizeof(hip.AdamwTensor))
            ta = time.perf_counter()
            assert L.msclip_adamw_multi(sub, min(4, self.n - i), *a, st) == 0
            worst = max(worst, time.perf_counter() - ta)
        T["run split: total"] = T.get("run split: total", 0.0) + time.perf_counter() - t0
        T["run split: slowest call"] = T.get("run split: slowest call", 0.0) + worst
        return
    t0 = time.perf_counter()
    orig_run(self, *a)
    T["AdamwPlan.run"] = T.get("AdamwPlan.run", 0.0) + time.perf_counter() - t0


hip.AdamwPlan.run = run
wrap(ts, "_adamw_plan", "_adamw_plan")
wrap(ts.eng, "repack_after_optimizer", "repack_after_optimizer")
for _ in range(4):
    ts.forward(img, tok); ts.step(ts.backward())
torch.cuda.synchronize()
T.clear()
n = 8
t00 = time.perf_counter()
tf = tb = tsx = 0.0
for _ in range(n):
    t0 = time.perf_counter(); ts.forward(img, tok)
    t1 = time.perf_counter(); g = ts.backward()
    t2 = time.perf_counter(); ts.step(g)
    t3 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1; tsx += t3 - t2
ti = time.perf_counter() - t00
torch.cuda.synchronize()
ta = time.perf_counter() - t00
print(f"[{mode}] host ms/step: forward {1e3 * tf / n:.2f} backward {1e3 * tb / n:.2f} step {1e3 * tsx / n:.2f}; issue {1e3 * ti / n:.1f}, wall {1e3 * ta / n:.1f}")
for k, v in T.items():
    print(f"   {k:30s} {1e3 * v / n:7.3f} ms per step")
