"""Worker of tests/test_gpu_model.py::test_two_rank_*: one process per rank.  argv[2] = "nccl" (default; RCCL, one GPU per
rank) or "gloo" (both ranks on GPU 0: everything of the N > 1 path except RCCL itself runs on a one-GPU box -- rank-major
gathers issued from the side stream, label offsets, the sharded loss with its scalar all-reduce, the bucketed gradient
all-reduce of the training step)."""
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
    backend = sys.argv[2] if len(sys.argv) > 2 else "nccl"
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    from conftest import synth_sd
    from msclip_amd import comm as C, synth
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    if world == 1:
        os.environ["MSCLIP_COLLECTIVES_AT_WORLD_1"] = "1"             # one rank, but every collective goes through the backend
    C.init_distributed(backend)
    assert dist.get_backend() == backend and C.comm.world_size == world
    name = "b32-yfcc-msclips"
    m = get_clip_model(named_config(name))
    m.load_state_dict(synth_sd(name), strict=True)
    m = m.cuda().eval()
    B = 6
    img, tok = synth.synth_images(B * world, seed=91), synth.synth_tokens(B * world, seed=92)
    mine = slice(rank * B, (rank + 1) * B)
    logits = m(img[mine].cuda(), tok[mine].cuda())                    # gather=True from the yaml (GATHER_TENSORS)
    loss = m.contrastive_loss(img[mine].cuda(), tok[mine].cuda())
    # SURVEY s8(e)'s single packed [B, 2, E] gather (behind a flag) against the default two per-modality gathers
    eng_ = m.engine()
    eng_.opt = eng_.opt.replace(gather_packed=True)
    logits_pk = m(img[mine].cuda(), tok[mine].cuda())
    loss_pk = m.contrastive_loss(img[mine].cuda(), tok[mine].cuda())
    assert m.engine()._ws[(B, B, "inference") if (B, B, "inference") in m.engine()._ws else (B, B)].get("pk") is not None
    eng_.opt = eng_.opt.replace(gather_packed=False)
    logits_back = m(img[mine].cuda(), tok[mine].cuda())               # ... and back to the dense buffers
    assert torch.equal(logits_back, logits)
    # the same collectives through the library's C ABI (msclip_comm_init / msclip_allgather_feats / msclip_allreduce: RCCL on the
    # compute stream, unique id carried by the process group's store) -- bitwise the ProcessGroupNCCL path, at the small batch
    # (eager launch loop) and at batch 256 per rank, where the step is a launch-plan replay that CONTAINS the two gathers
    native = {}
    if backend == "nccl":
        Bn = 256
        imgn, tokn = synth.synth_images(Bn, seed=93 + rank).cuda(), synth.synth_tokens(Bn, seed=94 + rank).cuda()
        pg_logits = m(imgn, tokn).clone()
        pg_loss = m.contrastive_loss(imgn, tokn).clone()
        C.init_native_comm(torch.cuda.current_device())
        eng_.opt = eng_.opt.replace(native_collectives=True)
        assert eng_._native_collectives()
        native["logits"] = m(img[mine].cuda(), tok[mine].cuda()).cpu()
        native["loss"] = float(m.contrastive_loss(img[mine].cuda(), tok[mine].cuda()))
        for _ in range(3):                                             # recording pass, then replays
            nat_logits = m(imgn, tokn)
            nat_loss = m.contrastive_loss(imgn, tokn)
        plan = eng_.last_plan
        native["planned"] = plan is not None and any(n == "msclip_allgather_feats" for n in plan.op_names())
        native["equal256"] = bool(torch.equal(nat_logits, pg_logits)) and bool(torch.equal(nat_loss, pg_loss))
        eng_.opt = eng_.opt.replace(native_collectives=False)
    # training step: gradients averaged over the ranks through comm.GradReducer (small buckets: several collectives)
    from msclip_amd import train
    ts = train.TrainStep(m, lr=1e-4, bn="frozen")
    tl = ts.forward(img[mine].cuda(), tok[mine].cuda())
    grads = ts.backward(bucket_bytes=8 << 20)
    keys = ["visual.transformer.resblocks.5.mlp.c_fc.weight", "visual.proj", "logit_scale", "ln_final.weight",
            "visual.transformer.parallel_branch_v.2.resnet_stage.conv_0.conv2.weight", "token_embedding.weight"]
    kept = {k: grads[k].float().cpu() for k in keys}
    n_grads, launched = len(grads), ts.reducer.launched
    ts.step(grads)                                                     # AdamW on the reducer's views (fresh: same generation)
    tl2 = ts.forward(img[mine].cuda(), tok[mine].cuda())               # the updated weights through the re-packed engine
    ts.backward()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"logits": logits.cpu(), "loss": float(loss), "logits_packed_gather": logits_pk.cpu(), "loss_packed_gather": float(loss_pk), "train_loss": float(tl), "launched": launched,
                    "grads": kept, "n_grads": n_grads, "train_loss_after_step": float(tl2), "native": native}, out)
    dist.barrier()
    C.destroy_native_comm()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
