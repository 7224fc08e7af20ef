"""Worker of tests/test_gpu_model.py::test_two_rank_rccl_*: one process per GPU, backend nccl (= RCCL)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
    torch.cuda.set_device(rank)
    from conftest import synth_sd
    from msclip_amd import comm as C, synth
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    C.init_distributed("nccl")
    assert dist.get_backend() == "nccl" and C.comm.world_size == world
    name = "b32-yfcc-msclips"
    m = get_clip_model(named_config(name))
    m.load_state_dict(synth_sd(name), strict=True)
    m = m.cuda().eval()
    B = 6
    img, tok = synth.synth_images(B * world, seed=91), synth.synth_tokens(B * world, seed=92)
    mine = slice(rank * B, (rank + 1) * B)
    logits = m(img[mine].cuda(), tok[mine].cuda())                    # gather=True from the yaml (GATHER_TENSORS)
    loss = m.contrastive_loss(img[mine].cuda(), tok[mine].cuda())
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"logits": logits.cpu(), "loss": float(loss)}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
