"""Same-box alternating A/B of forward-step schedules (C2: b32, batch 512): ms per step for each variant, `rounds` times in turn.
    python tools/probes/side_stream_ab.py [--cus 0,32,48,64] [--steps 30] [--rounds 3]
Variants: conv side stream on an ordinary stream (0) or confined to N CUs (EngineOptions.side_cu_mask = N); `inline` = no side
streams; `nt2` = captions staged ahead so that the text attention takes its 2-tile instantiation (longest caption <= 64)."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cus", default="0,32,48,64,96")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--model", default="b32-yfcc-msclips")
    a = ap.parse_args()
    from bench import load_schema
    from msclip_amd import synth
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    m = get_clip_model(named_config(a.model))
    m.load_state_dict(synth.synth_state_dict(load_schema(a.model), seed=0), strict=True)
    m = m.cuda().eval()
    eng = m.engine()
    base = eng.opt
    img, tok = synth.synth_images(a.batch, seed=10).cuda(), synth.synth_tokens(a.batch, seed=100).cuda()
    variants = {f"mask{c}": (base.replace(side_cu_mask=int(c)), False) for c in a.cus.split(",")}
    variants["low_priority_side"] = (base.replace(side_priority="low"), False)
    variants["no_text0_stream"] = (base.replace(text0_stream=False), False)
    variants["inline"] = (base.replace(conv_side_stream=False), False)
    variants["eager"] = (base.replace(plan=False), False)
    res = {k: [] for k in variants}
    for r in range(a.rounds):
        for k, (opt, staged) in variants.items():
            eng.opt = opt

            def step():
                if staged:
                    cap = eng.stage_captions(tok)
                    cap.totals()
                    return eng.forward_loss(img, cap, gather=True)
                return eng.forward_loss(img, tok, gather=True)
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            res[k].append(round((time.perf_counter() - t0) / a.steps * 1e3, 3))
    print(json.dumps({"model": a.model, "batch": a.batch, "steps": a.steps, "ms_per_step": res}))


if __name__ == "__main__":
    main()
