"""Where the HOST spends a forward step (or a training step): K steps issued against a GPU queue that is kept non-empty, so the
times are issue costs (launch-table replay, or ctypes calls + torch bookkeeping for the eager loop), not waits; then cProfile.
    python tools/probes/host_profile.py [--train --bn frozen] [--steps 10] [--top 30] [--no-plan]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--bn", default="frozen")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--sort", default="tottime")
    ap.add_argument("--no-plan", action="store_true", help="the eager launch loop (EngineOptions.plan = False)")
    ap.add_argument("--host-rows", action="store_true", help="round-5 path: the host reads each batch's row total (dynamic_rows = False)")
    a = ap.parse_args()
    from bench import load_schema
    from msclip_amd import synth, train
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    name = "b32-yfcc-msclips"
    m = get_clip_model(named_config(name))
    m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True)
    m = m.cuda().eval()
    eng = m.engine()
    eng.opt = eng.opt.replace(plan=not a.no_plan, dynamic_rows=not a.host_rows)
    img, tok = synth.synth_images(a.batch, seed=10).cuda(), synth.synth_tokens(a.batch, seed=100).cuda()
    ts = train.from_config(m, named_config(name), bn=a.bn) if a.train else None

    staged = {"cap": None}

    def step():
        if ts is None:
            return eng.forward_loss(img, tok, gather=True)
        # the training step reads each batch's row total on the host: staged one step ahead like bench.py (an input pipeline's
        # prefetch stage), so that what is timed is issue cost, not the wait for the read-back
        cap = staged["cap"] or eng.stage_captions(tok)
        staged["cap"] = eng.stage_captions(tok)
        loss = ts.forward(img, cap)
        ts.step(ts.backward())
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    kind = "eager launch loop" if (a.no_plan or ts is not None) else "launch-table replay"
    print(f"{kind}, {'host-read' if a.host_rows else 'device-side'} row counts: host issued {a.steps} steps in "
          f"{t_issue / a.steps * 1e3:.2f} ms/step, GPU finished them in {t_all / a.steps * 1e3:.2f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats(a.sort).print_stats(a.top)
    print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
