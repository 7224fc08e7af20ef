"""Which Python lines issue the large ATen kernels of a training step (torch.profiler with stacks): top device-time ATen ops."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from bench import load_schema
    from msclip_amd import synth, train
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    bn = sys.argv[1] if len(sys.argv) > 1 else "batch"
    name, B = "b32-yfcc-msclips", 512
    m = get_clip_model(named_config(name))
    m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True)
    m = m.cuda().eval()
    ts = train.TrainStep(m, lr=1e-4, bn=bn)
    img = synth.synth_images(B, seed=10).cuda()
    tok = synth.synth_tokens(B, seed=100).cuda()
    def one():
        ts.forward(img, tok)
        ts.step(ts.backward())
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    import collections
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode

    sites = collections.defaultdict(lambda: [0, 0])

    class Tap(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = func.__name__ if hasattr(func, "__name__") else str(func)
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if t is not None and t.is_cuda and not name.startswith(("view", "as_strided", "slice", "select", "t.", "transpose", "permute", "expand",
                                                                   "detach", "alias", "_unsafe_view", "unsqueeze", "squeeze", "reshape", "empty",
                                                                   "_reshape_alias", "split", "unbind", "narrow", "record_stream")):
                fr = [f for f in traceback.extract_stack() if "msclip_amd" in f.filename]
                where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "?"
                k = (name, where)
                sites[k][0] += t.numel() * t.element_size()
                sites[k][1] += 1
            return out

    with Tap():
        one()
    torch.cuda.synchronize()
    rows = sorted(sites.items(), key=lambda kv: -kv[1][0])
    tot = sum(v[0] for _, v in rows)
    print(f"ATen ops on device tensors in one step: {sum(v[1] for _, v in rows)} calls, {tot / 1e6:.1f} MB of outputs")
    for (name, where), (nbytes, n) in rows[:45]:
        print(f"{nbytes / 1e6:10.1f} MB x{n:3d}  {name:28s} {where}")


if __name__ == "__main__":
    main()
