"""Feasibility probe: the WHOLE training step (forward + backward + AdamW: three streams, ~900 launches, ~200 ATen ops, ~900 torch
allocations) captured into one hipGraph by torch.cuda.graph and replayed, against the eager step from the same state.  Full caption
rows (no host read in the step)."""
import os
import sys
import time

os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from bench import load_schema
    from msclip_amd import synth, train
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    name, B = "b32-yfcc-msclips", int(os.environ.get("PROBE_B", "512"))
    bn = os.environ.get("PROBE_BN", "frozen")
    img, tok = synth.synth_images(B, seed=10).cuda(), synth.synth_tokens(B, seed=100).cuda()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(s)

    def fresh():
        m = get_clip_model(named_config(name))
        m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True)
        m = m.cuda().eval()
        m.engine().opt = m.engine().opt.replace(text_pack=False)
        return m, train.TrainStep(m, lr=1e-5, bn=bn)
    # A: three eager steps.  B: two eager steps, then the third as a graph replay
    mA, tsA = fresh()
    lossesA = []
    for _ in range(3):
        l = tsA.forward(img, tok)
        tsA.step(tsA.backward())
        lossesA.append(float(l))
    ref = {k: v.detach().clone() for k, v in mA.state_dict().items()}
    del mA, tsA
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    m, ts = fresh()
    lossesB = []
    for _ in range(2):
        l = ts.forward(img, tok)
        ts.step(ts.backward())
        lossesB.append(float(l))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(g, stream=s):
        loss_g = ts.forward(img, tok)
        ts.step(ts.backward())
    print(f"captured in {time.perf_counter() - t0:.3f} s", flush=True)
    g.replay()
    torch.cuda.synchronize()
    lossesB.append(float(loss_g))
    print("eager losses", lossesA, "| eager, eager, graph:", lossesB, flush=True)
    dev = sorted(((ref[k].float() - v.float()).abs().max().item() / (ref[k].float().abs().max().item() + 1e-12), k)
                 for k, v in m.state_dict().items() if v.is_floating_point())
    nan = [k for k, v in m.state_dict().items() if v.is_floating_point() and not torch.isfinite(v).all()]
    print("parameters after step 3, graph vs eager: worst relative deviations", dev[-3:], "| bitwise equal tensors",
          sum(1 for d, _ in dev if d == 0.0), "of", len(dev), "| non-finite:", nan[:5], flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("loss after 3 more replays", float(loss_g), flush=True)
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"replay: host {t_issue / 10 * 1e3:.2f} ms/step, GPU {t_all / 10 * 1e3:.2f} ms/step, loss {float(loss_g):.5f}", flush=True)
    t0 = time.perf_counter()
    for _ in range(10):
        le = ts.forward(img, tok)
        ts.step(ts.backward())
    torch.cuda.synchronize()
    print(f"eager: {((time.perf_counter() - t0) / 10) * 1e3:.2f} ms/step, loss {float(le):.5f}", flush=True)


if __name__ == "__main__":
    main()
