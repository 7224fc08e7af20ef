"""Upper bound of what the lane stream's bias column sums cost the training step: A/B of the step as shipped against the step
with the bf16 column sums (c_fc / in_proj bias gradients: a re-read of dh [M, 4D] and dqkv [M, 3D] per layer on the lane
stream, beside the main queue's GEMMs) replaced by zeros -- WRONG gradients, a timing probe only.
    python tools/probes/lane_colsum_ab.py [--bn frozen] [--steps 12] [--rounds 4]"""
import argparse
import os
import sys
import time

os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bn", default="frozen")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--batch", type=int, default=512)
    a = ap.parse_args()
    from bench import load_schema
    from msclip_amd import hip, synth, train
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    name = "b32-yfcc-msclips"
    m = get_clip_model(named_config(name))
    m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True)
    m = m.cuda().eval()
    img, tok = synth.synth_images(a.batch, seed=10).cuda(), synth.synth_tokens(a.batch, seed=100).cuda()
    ts = train.from_config(m, named_config(name), bn=a.bn)
    real = hip.colsum
    zeros = {}

    def skipping(x, *args, **kw):
        if x.dtype == torch.bfloat16 and not args and not kw:
            z = zeros.get(x.shape[1])
            if z is None:
                z = zeros[x.shape[1]] = torch.zeros(x.shape[1], dtype=torch.float32, device=x.device)
            return z.clone()
        return real(x, *args, **kw)

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ts.forward(img, tok)
            ts.step(ts.backward())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    run(4)
    for r in range(a.rounds):
        hip.colsum = real
        run(2)
        ta = run(a.steps)
        hip.colsum = skipping
        run(2)
        tb = run(a.steps)
        print(f"round {r}: as shipped {ta:.3f} ms/step, bf16 column sums skipped {tb:.3f} ms/step")
    hip.colsum = real


if __name__ == "__main__":
    main()
