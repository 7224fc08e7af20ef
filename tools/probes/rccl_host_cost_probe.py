"""Host cost of launch-side primitives before and after an RCCL communicator exists in the process."""
import os
import sys
import time
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from msclip_amd import hip                                            # noqa: E402
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29681")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
x = torch.randn(64, 64, device=dev)
side = torch.cuda.Stream()
bg = hip.background_stream(dev)


def bench(tag, fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"  {tag:34s} host {1e6 * (t1 - t0) / n:7.2f} us/op   drained {1e6 * (t2 - t0) / n:7.2f} us/op", flush=True)


def ev_cross(st):
    def f():
        e = torch.cuda.Event()
        e.record(torch.cuda.current_stream())
        st.wait_event(e)
        with torch.cuda.stream(st):
            x.add_(1.0)
        torch.cuda.current_stream().wait_stream(st)
    return f


def alloc():
    t = torch.empty(1 << 18, device=dev)
    t.record_stream(side)
    del t


def suite(tag):
    print(tag, flush=True)
    bench("torch x.add_", lambda: x.add_(1.0))
    bench("ctypes kernel (cast_bf16)", lambda: hip.cast_bf16(x))
    bench("event + cross-stream op (torch)", ev_cross(side))
    bench("event + cross-stream op (low prio)", ev_cross(bg))
    bench("alloc + record_stream + free", alloc)


suite("before any process group")
dist.init_process_group("nccl")
y = torch.ones(4, device=dev)
dist.all_reduce(y)
torch.cuda.synchronize()
suite("with an RCCL communicator")
dist.destroy_process_group()
suite("after destroy_process_group")
