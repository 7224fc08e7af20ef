"""What creating an RCCL communicator does to the calling process: CPU affinity of the main thread, thread count."""
import os
import threading
import torch
import torch.distributed as dist
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29679")
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
def state(tag):
    aff = os.sched_getaffinity(0)
    nthr = len(os.listdir("/proc/self/task"))
    print(tag, "affinity", len(aff), sorted(aff)[:4], "...", "threads", nthr, "env", {k: v for k, v in os.environ.items() if "HIP" in k or "HSA" in k or "NCCL" in k or "RCCL" in k or "OMP" in k}, flush=True)
state("before")
dist.init_process_group("nccl")
state("after init")
x = torch.ones(4, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
state("after first collective")
import time
time.sleep(1.0)
# which threads burn CPU?
def cpu(tid):
    f = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)[1].split()
    return int(f[11]) + int(f[12])
t0 = {t: cpu(t) for t in os.listdir("/proc/self/task")}
time.sleep(2.0)
for t in os.listdir("/proc/self/task"):
    d = cpu(t) - t0.get(t, 0)
    if d > 5:
        print("busy thread", t, open(f"/proc/self/task/{t}/comm").read().strip(), d, "ticks in 2 s", flush=True)
dist.destroy_process_group()
