"""Where the gradient reducer's time goes (one-rank RCCL group, every collective an identity): times the ViT-B/32 batch-512
training step with (a) no reducer, (b) the default 64 MiB buckets, (c) one bucket at the end."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29677",
                  MSCLIP_COLLECTIVES_AT_WORLD_1="1")
from msclip_amd import comm as C, synth, train                      # noqa: E402
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model         # noqa: E402
from msclip_amd.config import named_config                          # noqa: E402
import json                                                         # noqa: E402


def load_schema(name):
    with open(os.path.join(ROOT, "tests", "golden", name + ".schema.json")) as f:
        return [(k, tuple(s), getattr(torch, d)) for k, s, d in json.load(f)]


torch.cuda.set_device(0)
MODE = sys.argv[2] if len(sys.argv) > 2 else ""
if MODE == "nopg":
    os.environ["MSCLIP_COLLECTIVES_AT_WORLD_1"] = "0"
elif MODE == "gloo":
    C.init_distributed("gloo")
else:
    C.init_distributed("nccl")
print("cpus", os.cpu_count(), "mode", MODE, flush=True)
name = "b32-yfcc-msclips"
m = get_clip_model(named_config(name))
m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True)
m = m.cuda().eval()
B = int(os.environ.get("PROBE_BATCH", "512"))
img, tok = synth.synth_images(B, seed=1).cuda(), synth.synth_tokens(B, seed=2).cuda()
ts = train.from_config(m, named_config(name), bn=sys.argv[1] if len(sys.argv) > 1 else "frozen")


def host_times(label, **kw):
    """Host-side return time of each call (no synchronisation in between): a call that blocks the host shows up here."""
    torch.cuda.synchronize()
    acc = [0.0, 0.0, 0.0]
    for _ in range(4):
        t0 = time.perf_counter(); ts.forward(img, tok)
        t1 = time.perf_counter(); g = ts.backward(**kw)
        t2 = time.perf_counter(); ts.step(g)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        acc = [acc[0] + t1 - t0, acc[1] + t2 - t1, acc[2] + t3 - t2]
    print(f"{label:40s} host returns after: forward {acc[0] / 4 * 1e3:.1f}  backward {acc[1] / 4 * 1e3:.1f}  step {acc[2] / 4 * 1e3:.1f} ms", flush=True)


def run(label, **kw):
    for _ in range(3):
        ts.forward(img, tok); ts.step(ts.backward(**kw))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        ts.forward(img, tok); ts.step(ts.backward(**kw))
    torch.cuda.synchronize()
    print(f"{label:40s} {(time.perf_counter() - t0) / 8 * 1e3:8.2f} ms", flush=True)


if MODE == "pgonly":          # communicator exists, but the data path issues no collective
    x = torch.ones(4, device="cuda")
    dist.all_reduce(x)
    os.environ["MSCLIP_COLLECTIVES_AT_WORLD_1"] = "0"
    assert not C.comm.collectives
if os.environ.get("PROBE_CPROFILE"):
    import cProfile
    import pstats
    for _ in range(3):
        ts.forward(img, tok); ts.step(ts.backward(reduce=False))
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        ts.forward(img, tok); ts.step(ts.backward(reduce=False))
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    sys.exit(0)
run("reduce=False", reduce=False)
if os.environ.get("PROBE_OWN_STREAM"):
    from msclip_amd import hip
    own = hip.compute_stream(torch.device('cuda', 0)) if os.environ.get('PROBE_OWN_STREAM') == 'high' else torch.cuda.Stream()
    own.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(own):
        run("reduce=False, on a non-default stream", reduce=False)
        host_times("  host", reduce=False)
    torch.cuda.current_stream().wait_stream(own)
    sys.exit(0)
run("64 MiB buckets", bucket_bytes=64 << 20)
run("one bucket (4 GiB)", bucket_bytes=4 << 30)
run("16 MiB buckets", bucket_bytes=16 << 20)
run("reduce=False again", reduce=False)
if dist.is_initialized():
    dist.destroy_process_group()
