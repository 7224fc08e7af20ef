"""Is the lane-stream interaction seen under a hipGraph replay (profiles/r06_train_hipgraph_probe.txt) a latent race of the eager
training step?  Delay every lane-stream job (a spin kernel in front of it) and, separately, the main stream behind every lane hand-off,
and compare all gradients with the undisturbed step: a true hazard (a buffer reused or read before its producer / after its
consumer on the other stream) shows up as a difference; correct event edges make the results independent of timing."""
import os
import sys

os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from bench import load_schema
    from msclip_amd import gradgemm, synth, train
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    name, B = "b32-yfcc-msclips", 256
    m = get_clip_model(named_config(name))
    m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True)
    m = m.cuda().eval()
    img, tok = synth.synth_images(B, seed=10).cuda(), synth.synth_tokens(B, seed=100).cuda()
    for bn in ("frozen", "batch"):
        ts = train.TrainStep(m, lr=1e-5, bn=bn)

        def grads():
            ts.forward(img, tok)
            g = {k: v.float().clone() for k, v in ts.backward().items()}
            torch.cuda.synchronize()
            return g
        ref = grads()
        again = grads()
        base = sum(1 for k in ref if not torch.equal(ref[k], again[k]))
        real_lane = gradgemm.lane
        spin = int(os.environ.get("PROBE_SPIN", "3000000"))

        def slow_lane(dev):                                   # every job that asks for the lane first queues a spin on it
            ln = real_lane(dev)
            with torch.cuda.stream(ln):
                torch.cuda._sleep(spin)
            return ln
        gradgemm.lane = slow_lane
        late = grads()
        gradgemm.lane = real_lane

        def slow_main(dev):                                   # ... or on the main stream, so that the lane runs ahead
            torch.cuda._sleep(spin // 4)
            return real_lane(dev)
        gradgemm.lane = slow_main
        early = grads()
        gradgemm.lane = real_lane
        for tag, g in (("lane delayed", late), ("main delayed", early)):
            bad = [k for k in ref if not torch.equal(ref[k], g[k]) and k != "token_embedding.weight"]
            print(f"bn={bn}: {tag}: {len(bad)} of {len(ref)} gradient tensors differ from the undisturbed step "
                  f"(run-to-run baseline: {base}; token_embedding.weight uses atomics) {bad[:6]}", flush=True)


if __name__ == "__main__":
    main()
