"""End-to-end images/s of the zero-shot evaluation loop (tools/eval_zeroshot.py -> msclip_amd.zeroshot.evaluate) on a GENERATED
ImageFolder, input pipeline on / off, next to the GPU-only encode rate (reference tools/zero_shot.py:70-81, 202-217, 253-275).

    python tools/eval_pipeline_bench.py [--images 10000] [--classes 100] [--workers 16] [--root /tmp/msclip_eval_imgs]

Images are synthetic JPEGs of ImageNet-like sizes (500 x 375, smooth random content so that the files compress like photos:
~40-60 KB); weights are random-init (the rate does not depend on them).  Prints one JSON record."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msclip_amd                                                   # noqa: E402

msclip_amd.configure_runtime()
import numpy as np                                                  # noqa: E402
import torch                                                        # noqa: E402


def make_folder(root, n_images, n_classes, seed=0):
    from PIL import Image
    from concurrent.futures import ThreadPoolExecutor
    per = n_images // n_classes
    done = os.path.join(root, f".done_{n_images}_{n_classes}")
    if os.path.exists(done):
        return
    rng = np.random.default_rng(seed)
    base = [rng.integers(0, 256, (24, 32, 3), dtype=np.uint8) for _ in range(64)]

    def one(args):
        c, i = args
        d = os.path.join(root, "val", f"n{c:08d}")
        os.makedirs(d, exist_ok=True)
        r = np.random.default_rng(c * 100003 + i)
        small = base[(c + i) % 64].astype(np.int16) + r.integers(-40, 40, (24, 32, 3))
        img = Image.fromarray(np.clip(small, 0, 255).astype(np.uint8)).resize((500, 375), Image.BICUBIC)
        img.save(os.path.join(d, f"img_{i:05d}.JPEG"), quality=90)
    with ThreadPoolExecutor(min(32, os.cpu_count() or 4)) as ex:
        list(ex.map(one, [(c, i) for c in range(n_classes) for i in range(per)]))
    open(done, "w").close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=10000)
    ap.add_argument("--classes", type=int, default=100)
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--root", default="/tmp/msclip_eval_imgs")
    ap.add_argument("--serial-images", type=int, default=1500, help="images of the single-threaded leg (it is slow)")
    a = ap.parse_args()
    from bench import load_schema
    from msclip_amd import synth, zeroshot
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    from msclip_amd.tokenizer import SimpleTokenizer
    t0 = time.perf_counter()
    make_folder(a.root, a.images, a.classes)
    t_make = time.perf_counter() - t0
    name = "b32-yfcc-msclips"
    m = get_clip_model(named_config(name))
    m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True)
    m = m.cuda().eval()
    classes, templates = zeroshot.load_prompts("imagenet")
    tok = SimpleTokenizer()
    val = os.path.join(a.root, "val")
    quiet = lambda *_: None
    # classifier once (not part of the image rate), reused by giving evaluate a model whose classifier build is cached
    rec = {"images": a.images, "classes": a.classes, "batch": a.batch, "host_cores": os.cpu_count(), "folder_build_s": round(t_make, 1)}
    # GPU-only: encode_image on a resident batch
    x = synth.synth_images(a.batch, seed=1).cuda()
    for _ in range(3):
        m.encode_image(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.encode_image(x)
    torch.cuda.synchronize()
    rec["gpu_only_images_per_s"] = round(20 * a.batch / (time.perf_counter() - t0), 1)
    zeroshot.evaluate(m, tok, val, classes, templates[:4], batch_size=a.batch, max_classes=a.classes, workers=a.workers, log=quiet)   # (page cache warm)
    rec["pipeline"] = []
    for wk in ([a.workers] if a.workers else [4, 16, 64]):
        res = zeroshot.evaluate(m, tok, val, classes, templates[:4], batch_size=a.batch, max_classes=a.classes, workers=wk, processes=0, log=quiet)
        rec["pipeline"].append({"images_per_s": round(res["images_per_s"], 1), "threads": res["loader_threads"], "n": res["n"], "top1": res["top1"]})
    for pr in [8, 16, 32, 64, 96]:
        if pr > (os.cpu_count() or 1):
            continue
        res = zeroshot.evaluate(m, tok, val, classes, templates[:4], batch_size=a.batch, max_classes=a.classes, processes=pr, log=quiet)
        rec["pipeline"].append({"images_per_s": round(res["images_per_s"], 1), "processes": res["loader_processes"], "n": res["n"], "top1": res["top1"],
                                "elapsed_s": round(res["elapsed_s"], 2), "consumer_s": {k: round(v, 2) for k, v in res["loader_stats"].items()}})
    res = zeroshot.evaluate(m, tok, val, classes, templates[:4], batch_size=a.batch, max_classes=a.classes, log=quiet)      # the default choice
    rec["default"] = {"images_per_s": round(res["images_per_s"], 1), "processes": res["loader_processes"], "threads": res["loader_threads"]}
    ser = zeroshot.evaluate(m, tok, val, classes, templates[:4], batch_size=a.batch, max_classes=a.classes, workers=0, log=quiet,
                            max_images=a.serial_images)
    rec["single_thread_loader"] = {"images_per_s": round(ser["images_per_s"], 1), "n": ser["n"]}
    rec["note"] = ("end-to-end rate includes file decode (JPEG 500x375), bicubic resize, crop, H2D copy, GPU normalisation, encode_image, "
                   "logits and top-k; the classifier build (text tower) is outside the timed part")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
