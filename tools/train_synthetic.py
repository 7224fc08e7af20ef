"""A few optimizer steps of the MS-CLIP-S training step on synthetic image / caption pairs (random-init weights):

    python tools/train_synthetic.py --model b32-yfcc-msclips --batch 64 --steps 20 [--bn batch|frozen] [--lr 2e-5]

Prints the contrastive loss of every step (the same fixed batches are cycled, so it has to fall), the step time and,
at the end, the inference-path loss of the first batch with the trained weights / running statistics.  One process per
GPU under torch.distributed.run for N > 1 (gradients are averaged over the ranks in 64 MiB buckets over RCCL)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import msclip_amd                                        # noqa: E402

msclip_amd.configure_runtime()                           # HSA_KERNARG_POOL_SIZE: before the first HIP call of the process
import torch                                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="b32-yfcc-msclips")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--nbatches", type=int, default=2, help="distinct synthetic batches cycled through")
    ap.add_argument("--bn", choices=("batch", "frozen"), default="batch")
    ap.add_argument("--lr", type=float, default=2e-5)
    ap.add_argument("--steps-per-epoch", type=int, default=0,
                    help="> 0: follow the yaml's TRAIN.LR_SCHEDULER (timm cosine + warm-up, train.CosineSchedule) with this many steps per epoch")
    args = ap.parse_args()
    from msclip_amd import comm as C, synth, train
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    C.init_distributed("nccl")
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if dist.is_initialized():                              # a process group exists: leave the legacy default stream (once, explicitly)
        from msclip_amd import hip
        hip.use_compute_stream(dev)
    cfg = named_config(args.model)
    from bench import load_schema
    model = get_clip_model(cfg)
    model.load_state_dict(synth.synth_state_dict(load_schema(args.model), seed=0), strict=True)
    model = model.to(dev).eval()
    ts = train.from_config(model, cfg, bn=args.bn)
    ts.lr = ts.lr_share = args.lr                          # (the schedule scales these base rates epoch by epoch)
    rank = C.comm.rank
    data = [(synth.synth_images(args.batch, seed=1000 * rank + 10 + i).to(dev),
             synth.synth_tokens(args.batch, seed=1000 * rank + 100 + i).to(dev)) for i in range(args.nbatches)]
    losses = []
    for step in range(args.steps):
        img, tok = data[step % args.nbatches]
        if args.steps_per_epoch > 0 and step % args.steps_per_epoch == 0:
            ts.set_epoch(step // args.steps_per_epoch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = ts.forward(img, tok)
        ts.step(ts.backward())
        torch.cuda.synchronize()
        losses.append(float(loss))
        if rank == 0:
            print(f"step {step:3d}  loss {losses[-1]:.4f}  {1e3 * (time.perf_counter() - t0):7.1f} ms", flush=True)
    # the inference-path loss gathers features and all-reduces its partial sums (GATHER_TENSORS: True): EVERY rank runs it,
    # rank 0 prints it
    inf = float(model.contrastive_loss(*data[0]))
    ok = all(l == l for l in losses) and min(losses[-args.nbatches:]) < losses[0]
    if rank == 0:
        print(f"first batch through the inference path (running statistics): loss {inf:.4f}")
        print("OK" if ok else "FAILED: the loss did not fall")
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
