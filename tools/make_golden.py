"""Generate tests/golden/ from the REAL reference (build container only).

Imports /root/reference through tools/ref_import.py, fills it with the
deterministic synthetic weights of msclip_amd.synth, runs it in fp32 on CPU
and stores inputs-by-seed / outputs-by-value as small .npz fixtures, plus the
state_dict schema (key order, shapes, dtypes) that is the checkpoint ABI.

    python tools/make_golden.py            # writes tests/golden/*.npz, *.json

Only data leaves this script: no reference source or bytecode is copied.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_import as R                      # noqa: E402
from msclip_amd import synth                # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 0
BATCH = 4


def summarize(t):
    """Compact, order-sensitive summary of a big activation: mean, abs-mean and a
    strided 64-element slice of the flattened tensor."""
    f = t.detach().float().flatten()
    idx = torch.linspace(0, f.numel() - 1, 64).long().clamp_(max=f.numel() - 1)   # (fp32 linspace overshoots past 2^24 elements)
    return np.concatenate([[f.mean().item(), f.abs().mean().item()], f[idx].numpy()]).astype(np.float32)


def build_only(name):
    """The reference module with the synthetic weights loaded, without writing the forward fixtures."""
    model, cfg = R.build_reference_model(name)
    sd = synth.synth_state_dict(synth.schema_of(model), seed=SEED)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return model, cfg


def run_config(name):
    model, cfg = R.build_reference_model(name)
    schema = synth.schema_of(model)
    sd = synth.synth_state_dict(schema, seed=SEED)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    # aliasing sanity: shared tensors must still be shared after the load
    v = model.visual.transformer.resblocks[3].attn.in_proj_weight
    t = model.transformer.resblocks[3].attn.in_proj_weight
    assert v.data_ptr() == t.data_ptr()

    img = synth.synth_images(BATCH, seed=SEED)
    tok = synth.synth_tokens(BATCH, seed=SEED + 1)

    taps = {}
    hooks = []
    vt = model.visual.transformer

    def tap(nm):
        def fn(_m, _i, o):
            taps[nm] = o[1] if isinstance(o, tuple) else o
        return fn

    hooks.append(vt.resblocks[0].register_forward_hook(tap("stem_out")))
    hooks.append(vt.resblocks[0].relu.register_forward_hook(tap("stem_conv1")))
    for i in range(4):
        hooks.append(vt.resblocks[0].resnet_stage[i].register_forward_hook(tap(f"stem_stage{i}")))
    hooks.append(model.visual.ln_pre.register_forward_hook(tap("tokens_ln_pre")))
    for j in range(5):
        hooks.append(vt.parallel_branch_v[j].register_forward_hook(tap(f"parallel{j}")))
        hooks.append(vt.parallel_lateral_adapter[j].register_forward_hook(tap(f"adapter{j}")))
    for i in (1, 2, 11):
        hooks.append(vt.resblocks[i].register_forward_hook(tap(f"vblock{i}")))
    for i in (0, 1, 11):
        hooks.append(model.transformer.resblocks[i].register_forward_hook(tap(f"tblock{i}")))

    with torch.no_grad():
        fi = model.encode_image(img)
        ft = model.encode_text(tok)
        fi_raw = model.encode_image(img, norm=False)
        ft_raw = model.encode_text(tok, norm=False)
        for h in hooks:
            h.remove()
        R.ensure_single_rank_group()
        logits = model(img, tok)

    out = {
        "seed": np.int64(SEED), "batch": np.int64(BATCH),
        "image_features": fi.numpy(), "text_features": ft.numpy(),
        "image_features_raw": fi_raw.numpy(), "text_features_raw": ft_raw.numpy(),
        "logits": logits.numpy(),
    }
    for k, v in taps.items():
        # reference activations are seq-first [L, B, C]; store batch-first
        if k.startswith(("vblock", "tblock", "adapter")):
            v = v.permute(1, 0, 2)
        out["tap_" + k] = summarize(v)
        out["tapshape_" + k] = np.array(v.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)

    with open(os.path.join(OUT, f"{name}.schema.json"), "w") as f:
        json.dump([[k, list(s), str(d).replace("torch.", "")] for k, s, d in schema], f, indent=0)
    n_unique = sum(p.numel() for p in model.parameters())
    print(f"{name}: {len(schema)} keys, {n_unique} unique params, logits diag {logits.diag().numpy()}")
    return model, sd


def c1_images(n_classes=8, per_class=8, seed=7):
    """The generated 64-image ImageFolder of BASELINE config C1: uint8 [n_classes, per_class, 224, 224, 3] from a seed
    (smooth random fields so that PNG round trips and the conv stem see image-like statistics)."""
    rng = np.random.default_rng(seed)
    low = rng.integers(0, 256, (n_classes, per_class, 14, 14, 3)).astype(np.float32)
    img = np.repeat(np.repeat(low, 16, axis=2), 16, axis=3)
    img += rng.normal(0, 12, img.shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def zeroshot_fixture(model, name):
    """Real-token text features and the C1 zero-shot protocol from the reference (tools/zero_shot.py:122-134, 265-266):
    features of the committed prompts of tests/golden/tokenizer.json, the classifier columns of the first 8 ImageNet
    classes over all 80 templates, and 100 * f_img @ W for the 64 generated images."""
    from dataset.languages.simple_tokenizer import SimpleTokenizer
    from msclip_amd import zeroshot
    tok = SimpleTokenizer()
    g = json.load(open(os.path.join(OUT, "tokenizer.json")))
    ids = torch.tensor(g["ids"], dtype=torch.long)
    assert tok(g["prompts"]).tolist() == g["ids"]
    classes, templates = zeroshot.load_prompts("imagenet")
    classes = classes[:8]
    imgs = c1_images()
    x = torch.stack([zeroshot.preprocess_array(a) for a in imgs.reshape(-1, 224, 224, 3)])
    with torch.no_grad():
        ft = model.encode_text(ids)
        cols = []
        for c in classes:                                                    # zero_shot.py:122-134
            e = model.encode_text(tok([t.format(c) for t in templates]))
            e = e / e.norm(dim=-1, keepdim=True)
            e = e.mean(dim=0)
            cols.append(e / e.norm())
        W = torch.stack(cols, dim=1)
        fi = torch.cat([model.encode_image(x[i:i + 16]) for i in range(0, x.shape[0], 16)])
        logits = 100.0 * fi @ W
    y = torch.arange(8).repeat_interleave(8)
    top1 = (logits.argmax(-1) == y).float().mean().item() * 100.0
    np.savez_compressed(os.path.join(OUT, f"{name}.zeroshot.npz"), prompt_text_features=ft.numpy(),
                        classifier=W.numpy(), image_features=fi.numpy(), logits=logits.numpy(),
                        top1=np.float32(top1), n_classes=np.int64(8), per_class=np.int64(8), image_seed=np.int64(7))
    print(f"{name}: zero-shot fixture top1 {top1:.2f}% on 64 generated images, W {tuple(W.shape)}")


def grads_fixture(model, name, train_bn=False, batch=None, tag=None, full_limit=4096):
    """f3 (backward) fixture: autograd of the REAL reference through forward(image, text) and the symmetric CE
    0.5 * (CE(logits) + CE(logits^T)) (the loss itself is not in the reference), eval-mode BatchNorm,
    fp32, the golden batch.  Stored per parameter: mean, abs-mean, abs-max and a 64-point strided sample of the gradient (the
    shared tensors' gradients are the SUM over both towers: one Parameter object, M.py:2808-2830), plus the loss."""
    nb = batch or BATCH                      # train-mode BN: 16 (statistics over 4 images are ill-conditioned in any precision)
    img = synth.synth_images(nb, seed=SEED)
    tok = synth.synth_tokens(nb, seed=SEED + 1)
    R.ensure_single_rank_group()
    for p in model.parameters():
        p.grad = None
        p.requires_grad_(True)
    if train_bn:
        # train(): every BatchNorm normalises with the batch statistics of this (per-GPU) batch and updates its running
        # statistics (momentum 0.1, unbiased variance); dropout / drop-path are 0 in the released configs
        model.train()
        before = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k}
    logits = model(img, tok)
    lab = torch.arange(nb)
    loss = 0.5 * (torch.nn.functional.cross_entropy(logits, lab) + torch.nn.functional.cross_entropy(logits.t(), lab))
    loss.backward()
    out = {"loss": np.float32(loss.item()), "batch": np.int64(nb), "seed": np.int64(SEED)}
    seen = {}
    for k, p in model.named_parameters(remove_duplicate=False):
        if p.grad is None:
            continue
        if id(p) in seen:                      # aliases (text-tower names of the shared tensors): same gradient object
            out["alias_" + k] = np.array(seen[id(p)])
            continue
        seen[id(p)] = k
        out["g_" + k] = summarize(p.grad)
        out["gmax_" + k] = np.float32(p.grad.abs().max().item())
        if p.grad.numel() <= full_limit:
            out["gfull_" + k] = p.grad.detach().numpy().astype(np.float32)
    if train_bn:
        after = model.state_dict()
        for k in before:                                   # the running statistics after ONE forward in train()
            out["run_" + k] = after[k].detach().numpy().astype(np.float32)
        model.load_state_dict({**after, **before})         # put the statistics back: later fixtures start from the same model
        model.eval()
    np.savez_compressed(os.path.join(OUT, f"{name}.grads{'_trainbn' if train_bn else ''}{tag or ''}.npz"), **out)
    print(f"{name}: grads fixture{' (train-mode BN)' if train_bn else ''}, loss {loss.item():.5f}, "
          f"{sum(k.startswith('g_') for k in out)} gradient tensors")
    for p in model.parameters():
        p.grad = None


def multirank_gather_fixture():
    """Reference gather_tensors under 2-rank gloo (lib/utils/comm.py:140-154):
    rank-major concat, and gradient only through the local slice."""
    import torch.multiprocessing as mp
    path = os.path.join(OUT, "gather_2rank.npz")
    mp.spawn(_gather_worker, args=(2, path), nprocs=2, join=True)


def _gather_worker(rank, world, path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29593"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R.import_reference_module()
    from utils.comm import gather_tensors
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(3, 8, generator=g, requires_grad=True)
    allx = gather_tensors(x)
    w = torch.arange(allx.numel(), dtype=torch.float32).reshape(allx.shape)
    (allx * w).sum().backward()
    if rank == 0:
        np.savez(path, gathered=allx.detach().numpy(), grad_rank0=x.grad.numpy(), x_rank0=x.detach().numpy())
    dist.barrier()
    dist.destroy_process_group()


def refresh_prompt_features(name="b32-yfcc-msclips"):
    """`--prompt-features-only`: re-capture ONLY prompt_text_features of <name>.zeroshot.npz from the reference for the
    current tests/golden/tokenizer.json (after prompts were added there); every other array of the file is kept."""
    model, _ = R.build_reference_model(name)
    model.load_state_dict(synth.synth_state_dict(synth.schema_of(model), seed=SEED), strict=True)
    model.eval()
    g = json.load(open(os.path.join(OUT, "tokenizer.json")))
    from dataset.languages.simple_tokenizer import SimpleTokenizer
    assert SimpleTokenizer()(g["prompts"]).tolist() == g["ids"]
    with torch.no_grad():
        ft = model.encode_text(torch.tensor(g["ids"], dtype=torch.long))
    path = os.path.join(OUT, f"{name}.zeroshot.npz")
    z = dict(np.load(path))
    n_old = z["prompt_text_features"].shape[0]
    assert np.abs(ft.numpy()[:n_old] - z["prompt_text_features"]).max() < 1e-5      # the old prompts still give the old rows
    z["prompt_text_features"] = ft.numpy()
    np.savez_compressed(path, **z)
    print(f"{name}: prompt_text_features {n_old} -> {ft.shape[0]} rows")


def l16_fixture(name="l16-fp8-msclips", batch=2):
    """`--l16`: BASELINE config C5's stand-in (experiments/model/l16-fp8-msclips.yaml) is an ordinary model for the
    reference (width 1024, 24 layers, 14 x 14 grid): build it THERE from that yaml, synthetic weights, and capture the
    state_dict schema, features and logits of a small batch -- the pin of this build's bf16 path for that model (the fp8
    path has no reference semantics)."""
    model, _ = R.build_reference_model(name)
    schema = synth.schema_of(model)
    model.load_state_dict(synth.synth_state_dict(schema, seed=SEED), strict=True)
    model.eval()
    with open(os.path.join(OUT, name + ".schema.json"), "w") as f:
        json.dump([(k, list(s), str(d).replace("torch.", "")) for k, s, d in schema], f)
    img, tok = synth.synth_images(batch, seed=SEED), synth.synth_tokens(batch, seed=SEED + 1)
    R.ensure_single_rank_group()
    taps, hooks = {}, []
    if hasattr(model.visual, "conv1"):          # the patch-conv stem (l14): tokens after ln_pre and a few blocks of both towers
        def tap(nm):
            def fn(_m, _i, o):
                taps[nm] = o[1] if isinstance(o, tuple) else o
            return fn
        hooks.append(model.visual.ln_pre.register_forward_hook(tap("tokens_ln_pre")))
        for i in (0, 1, 23):
            hooks.append(model.visual.transformer.resblocks[i].register_forward_hook(tap(f"vblock{i}")))
            hooks.append(model.transformer.resblocks[i].register_forward_hook(tap(f"tblock{i}")))
    with torch.no_grad():
        fi, ft = model.encode_image(img), model.encode_text(tok)
        for h in hooks:
            h.remove()
        fir = model.encode_image(img, norm=False)
        logits = model(img, tok)
    out = dict(image_features=fi.numpy(), text_features=ft.numpy(), image_features_raw=fir.numpy(), logits=logits.numpy(),
               batch=np.int64(batch), seed=np.int64(SEED))
    for k, v in taps.items():
        if k.startswith(("vblock", "tblock")):
            v = v.permute(1, 0, 2)              # reference activations are seq-first [L, B, C]; store batch-first
        out["tap_" + k] = summarize(v)
        out["tapshape_" + k] = np.array(v.shape, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    flops = None
    try:                                        # algorithmic forward FLOPs per pair, counted on the reference itself
        from torch.utils.flop_counter import FlopCounterMode
        with torch.no_grad(), FlopCounterMode(display=False) as fc:
            model.encode_image(img[:1])
        fi_flops = fc.get_total_flops()
        with torch.no_grad(), FlopCounterMode(display=False) as fc:
            model.encode_text(tok[:1])
        flops = (fi_flops / 1e9, fc.get_total_flops() / 1e9)
    except Exception as e:
        flops = repr(e)
    print(f"{name}: {len(schema)} keys, {sum(p.numel() for p in model.parameters())} unique params, features {tuple(fi.shape)}, "
          f"logits {logits.numpy().round(3).tolist()}, taps {sorted(taps)}, GFLOP/pair image, text = {flops}")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "--prompt-features-only" in sys.argv:
        return refresh_prompt_features()
    if "--grads-b32" in sys.argv:
        # round 6: the same reference-autograd fixtures at batch 32, where the cancelling sums behind the
        # conv-side and LayerNorm-bias gradients are well conditioned (the batch-4 fixture could not see a 10 % systematic error
        # there).  Sampled summaries + full copies of the <= 1024-element tensors keep each file under 2 MB.
        model, _ = build_only("b32-yfcc-msclips")
        grads_fixture(model, "b32-yfcc-msclips", batch=32, tag="_b32", full_limit=1024)
        grads_fixture(model, "b32-yfcc-msclips", train_bn=True, batch=32, tag="_b32", full_limit=1024)
        return
    if "--l16" in sys.argv:
        return l16_fixture()
    if "--l14" in sys.argv:                      # BASELINE config C5 proper: patch-14 conv stem, 16 x 16 grid (experiments/model/l14-fp8-msclips.yaml)
        return l16_fixture("l14-fp8-msclips")
    for name in ("b32-yfcc-msclips", "b16-yfcc-msclips"):
        model, _ = run_config(name)
        if name.startswith("b32"):
            zeroshot_fixture(model, name)
        grads_fixture(model, name)              # b16: the 197-token grid (query-blocked attention backward, k = 8 adapters)
        grads_fixture(model, name, train_bn=True, batch=16 if name.startswith("b32") else 8)
    multirank_gather_fixture()


if __name__ == "__main__":
    main()
