"""Zero-shot classification driver (reference tools/zero_shot.py:122-134, 149-163, 183-310).

* classifier: per class, tokenize every prompt template, encode_text, mean over templates, renormalise, stack as
  columns -> W[embed, n_classes]                                         (zero_shot.py:122-134)
* images: Resize(224, bicubic) -> CenterCrop(224) -> ToTensor -> Normalize(ImageNet mean/std), done with PIL + numpy
  (torchvision is not installed; the reference uses lib/config/default.py:84-85 statistics)   (zero_shot.py:202-207)
* logits = 100 * f_img @ W, top-k accuracy                                (zero_shot.py:265-266, 149-163)
* ImageFolder layout val/<wnid>/*.JPEG, class index = sorted directory names (DATASET/DATA.md:5-14)

Class names and prompt templates are DATA: ImageNet's 1000 names and 80 templates (the values of the reference's
lib/dataset/prompts/constants.py) ship as msclip_amd/data/imagenet_prompts.json; `load_prompts` reads that file, another
json of the same layout, or a python constants file.
"""
import json
import os
import runpy

import numpy as np
import torch

from . import hip

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp", ".ppm", ".webp")


PROMPTS = {"imagenet": os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "imagenet_prompts.json")}


TRANSFER_NAME = {"oxford-flower-102": "flower102-tf", "fgvc-aircraft-2013b": "fgvc-aircraft-2013b-variants102"}


def prompt_name(dataset):
    """zero_shot.py:235-238."""
    return TRANSFER_NAME.get(dataset, dataset)


def load_prompts(path="imagenet"):
    """-> (classnames, templates).  `path`: a dataset name with packaged prompts ("imagenet", as config.DATASET.DATASET
    selects them in zero_shot.py:235-243), a json {"classes": [...], "templates": [...]} or a python constants file
    defining IMAGENET_CLASSES / IMAGENET_DEFAULT_TEMPLATES (the reference's constants.py layout)."""
    if path in PROMPTS:
        path = PROMPTS[path]
    elif not os.path.exists(path):
        raise ValueError("Can not find prompt for dataset: {}".format(path))        # zero_shot.py:243
    if path.endswith(".json"):
        with open(path) as f:
            d = json.load(f)
        return list(d["classes"]), list(d["templates"])
    ns = runpy.run_path(path)
    classes = ns.get("IMAGENET_CLASSES") or ns["ALL_CLASSES_DICT"]["imagenet"]
    templates = ns.get("IMAGENET_DEFAULT_TEMPLATES") or ns["ALL_TEMPLATES_DICT"]["imagenet"]
    return list(classes), [t if isinstance(t, str) else t("{}") for t in templates]


@torch.no_grad()
def zeroshot_classifier(model, tokenizer, classnames, templates, device="cuda", classes_per_batch=8):
    """W[embed, n_classes]; several classes are encoded per encode_text call (the reference does one class at a time)."""
    cols = []
    n_t = len(templates)
    for c0 in range(0, len(classnames), classes_per_batch):
        group = classnames[c0:c0 + classes_per_batch]
        texts = [t.format(c) for c in group for t in templates]
        emb = model.encode_text(tokenizer(texts).to(device)).float()          # unit rows
        emb = emb.reshape(len(group), n_t, -1).mean(dim=1)
        cols.append(emb / emb.norm(dim=-1, keepdim=True))
    return torch.cat(cols, 0).t().contiguous()


def accuracy(output, target, topk=(1,)):
    """Top-k hits in percent (zero_shot.py:149-163)."""
    maxk = max(topk)
    pred = output.topk(maxk, 1, True, True)[1].t()
    correct = pred.eq(target.reshape(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum().item() * 100.0 / target.shape[0] for k in topk]


def preprocess(img, size=224, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """PIL image -> float32 [3, size, size]: shorter side to `size` (bicubic), centre crop, /255, normalise."""
    from PIL import Image
    img = img.convert("RGB")
    w, h = img.size
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    img = img.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    img = img.crop((left, top, left + size, top + size))
    x = np.asarray(img, dtype=np.float32) / 255.0
    x = (x - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))


from ._decode_worker import load_pixels            # noqa: E402  (decode + bicubic resize + centre crop -> uint8 [size, size, 3])


def effective_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a container with 256 visible
    cores and `cpu.max = 1600000 100000` has 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def normalise_lut(mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """float32 [3, 256]: (v / 255 - mean_c) / std_c for every pixel value v, computed with the SAME float32 numpy arithmetic as
    `preprocess`, so that looking pixels up in it on the GPU is bit-identical to normalising them on the host."""
    v = np.arange(256, dtype=np.float32) / 255.0
    return torch.from_numpy(np.stack([(v - np.float32(m)) / np.float32(s) for m, s in zip(mean, std)]).astype(np.float32))


class ImagePipeline:
    """The input pipeline of the zero-shot loop (reference tools/zero_shot.py:70-81, 202-217, 262: a 6-worker DataLoader with
    pinned memory and non_blocking copies).  A pool of `workers` threads decodes / resizes / crops files to uint8 pixels
    (load_pixels) straight into one of `depth` pinned staging batches; a batch that is complete goes to the device on a copy
    stream (150 KB per image instead of the 600 KB of fp32 pixels) and is normalised there by table look-up (normalise_lut:
    bit-identical to the host arithmetic) into the NCHW fp32 tensor encode_image takes.  Iterating yields (images, labels,
    n) with the next batches' decoding and copies already in flight."""

    def __init__(self, items, batch_size, device, size=224, mean=IMAGENET_MEAN, std=IMAGENET_STD, workers=None, depth=3, processes=0):
        import concurrent.futures as cf
        self.items, self.bs, self.dev, self.size = list(items), int(batch_size), torch.device(device), size
        self.workers = max(1, min(16, effective_cores()) if workers is None else int(workers))
        self.depth = max(2, depth)
        self.processes = int(processes or 0)
        self.pool = self.shm = None
        if self.processes > 0:
            # decoding PROCESSES (no GIL between them): plain subprocesses of msclip_amd._decode_worker (numpy + PIL only; neither
            # forks of this process with its live HIP runtime nor multiprocessing children that re-import __main__), writing into
            # one shared-memory staging area [depth, batch, size, size, 3]; a task = a run of 8 files of a batch, dealt round-robin
            import queue
            import subprocess
            import sys
            import threading
            from multiprocessing import shared_memory
            self.shm = shared_memory.SharedMemory(create=True, size=self.depth * self.bs * size * size * 3)
            self.shm_np = np.ndarray((self.depth, self.bs, size, size, 3), dtype=np.uint8, buffer=self.shm.buf)
            pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env = dict(os.environ, PYTHONPATH=pkg_parent + os.pathsep + os.environ.get("PYTHONPATH", ""))
            self.procs = [subprocess.Popen([sys.executable, "-u", "-m", "msclip_amd._decode_worker", self.shm.name, str(self.depth),
                                            str(self.bs), str(size)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, text=True,
                                           bufsize=1) for _ in range(self.processes)]
            self.done_q = queue.Queue()

            def reader(proc):
                for line in proc.stdout:
                    self.done_q.put(json.loads(line))
                self.done_q.put([-1, 0, "a decoding process exited"])
            self.readers = [threading.Thread(target=reader, args=(p,), daemon=True) for p in self.procs]
            for t in self.readers:
                t.start()
            self.pending, self.rr = {}, 0
        else:
            self.pool = cf.ThreadPoolExecutor(self.workers)
        self.pin = [torch.empty(self.bs, size, size, 3, dtype=torch.uint8).pin_memory() for _ in range(self.depth)]
        self.gpu = [torch.empty(self.bs, size, size, 3, dtype=torch.uint8, device=self.dev) for _ in range(self.depth)]
        self.free = [torch.cuda.Event() for _ in range(self.depth)]          # "the consumer is done with gpu[k]"
        self.lut = normalise_lut(mean, std).to(self.dev).reshape(-1)           # [3 * 256]
        self.offs = (torch.arange(3, device=self.dev) * 256).view(1, 1, 1, 3)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.copied = {}                                                       # slot -> event "the H2D copy out of pin[slot] is done"
        self.stats = {"wait_decode_s": 0.0, "stage_copy_s": 0.0, "submit_s": 0.0, "gpu_prep_s": 0.0}   # where the consumer thread's time goes

    def _decode_into(self, k, j, path):
        self.pin[k][j].copy_(torch.from_numpy(load_pixels(path, self.size)))

    def _submit(self, b):
        k = b % self.depth
        chunk = self.items[b * self.bs:(b + 1) * self.bs]
        if self.processes > 0:
            self.pending[b] = len(chunk)
            for j0 in range(0, len(chunk), 8):
                proc = self.procs[self.rr % self.processes]
                self.rr += 1
                proc.stdin.write(json.dumps([b, k, j0, [p for p, _ in chunk[j0:j0 + 8]]]) + "\n")
                proc.stdin.flush()
            return b, chunk
        return [self.pool.submit(self._decode_into, k, j, p) for j, (p, _) in enumerate(chunk)], chunk

    def _wait(self, futs, b):
        if self.processes == 0:
            for f in futs:
                f.result()                                   # (re-raises a decoding error with the worker's traceback)
            return
        while self.pending.get(b, 0) > 0:                    # completions of later batches are booked as they arrive
            try:
                bb, n, err = self.done_q.get(timeout=120)
            except Exception:
                raise RuntimeError("ImagePipeline: the decoding processes do not answer") from None
            if err is not None:
                raise RuntimeError("ImagePipeline: decoding failed in a worker process: " + err)
            self.pending[bb] = self.pending.get(bb, 0) - n
        del self.pending[b]
        import time
        t0 = time.perf_counter()
        k = b % self.depth
        n = min(self.bs, len(self.items) - b * self.bs)
        self.pin[k][:n].copy_(torch.from_numpy(self.shm_np[k, :n]))
        self.stats["stage_copy_s"] += time.perf_counter() - t0
        self.stats["wait_decode_s"] -= time.perf_counter() - t0

    def __iter__(self):
        nb = (len(self.items) + self.bs - 1) // self.bs
        inflight = {b: self._submit(b) for b in range(min(self.depth - 1, nb))}
        for b in range(nb):
            import time
            futs, chunk = inflight.pop(b)
            t0 = time.perf_counter()
            self._wait(futs, b)
            self.stats["wait_decode_s"] += time.perf_counter() - t0
            k, n = b % self.depth, len(chunk)
            cur = torch.cuda.current_stream(self.dev)
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(self.free[k])    # gpu[k]'s previous batch has been normalised
                self.gpu[k][:n].copy_(self.pin[k][:n], non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(self.copy_stream)
            self.copied[k] = landed
            nxt = b + self.depth - 1
            if nxt < nb:
                # pin[nxt % depth] was the staging area of batch b - 1, whose copy was queued an iteration ago: the HOST waits for
                # that copy before the workers overwrite the pinned buffer
                prev = self.copied.get(nxt % self.depth)
                if prev is not None and self.processes == 0:   # (threads write the pinned buffer itself; processes write shared memory)
                    prev.synchronize()
                t0 = time.perf_counter()
                inflight[nxt] = self._submit(nxt)
                self.stats["submit_s"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            cur.wait_event(landed)
            x = self.lut[(self.gpu[k][:n].long() + self.offs).reshape(-1)].view(n, self.size, self.size, 3)
            x = x.permute(0, 3, 1, 2).contiguous()
            self.free[k].record(cur)
            y = torch.tensor([c for _, c in chunk], device=self.dev)
            self.stats["gpu_prep_s"] += time.perf_counter() - t0
            yield x, y, n

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=True)
        if self.shm is not None:
            for p in self.procs:
                try:
                    p.stdin.close()                          # EOF ends the worker's loop
                except Exception:
                    pass
            for p in self.procs:
                try:
                    p.wait(timeout=10)
                except Exception:
                    p.kill()
            del self.shm_np
            self.shm.close()
            self.shm.unlink()
            self.shm = None


def preprocess_array(a, size=224, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """uint8 [H, W, 3] array -> the same tensor `preprocess` gives for the image file holding those pixels."""
    from PIL import Image
    return preprocess(Image.fromarray(np.asarray(a, dtype=np.uint8)), size, mean, std)


def image_folder(root):
    """[(path, class_index)], class directories sorted like torchvision's ImageFolder."""
    classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    items = []
    for ci, c in enumerate(classes):
        for dp, _, files in sorted(os.walk(os.path.join(root, c))):
            for f in sorted(files):
                if f.lower().endswith(IMG_EXT):
                    items.append((os.path.join(dp, f), ci))
    return classes, items


@torch.no_grad()
def evaluate(model, tokenizer, val_root, classnames, templates, batch_size=32, device="cuda", max_images=None,
             size=224, log=print, max_classes=None, mean=IMAGENET_MEAN, std=IMAGENET_STD, dataset="imagenet",
             metric="accuracy", return_logits=False, workers=None, processes=None):
    """Full zero-shot run: returns dict(top1, top5, n, images_per_s).  Logs the reference's final line (zero_shot.py:304-308).
    max_classes keeps the first C class directories and class names (subset runs); max_images a strided subset.
    workers: decoding threads of the input pipeline (ImagePipeline; None = one per host core up to 64; 0 = the single-threaded
    loop of rounds 1-5: PIL + numpy on the calling thread, blocking copies -- kept for the A/B and as the definition of the
    arithmetic)."""
    import time
    from PIL import Image
    hip.require_gpu()
    if metric != "accuracy":
        raise NotImplementedError(f"TEST.METRIC {metric!r}: only top-1 'accuracy' (ImageNet) is on the MS-CLIP-S eval path")
    dirs, items = image_folder(val_root)
    if max_classes:
        classnames = list(classnames)[:max_classes]
        items = [(p, c) for p, c in items if c < max_classes]
        dirs = dirs[:max_classes]
    if len(dirs) != len(classnames):
        raise ValueError(f"{len(dirs)} class directories under {val_root} but {len(classnames)} class names")
    if max_images:
        step = max(1, len(items) // max_images)
        items = items[::step][:max_images]
    W = zeroshot_classifier(model, tokenizer, classnames, templates, device)
    log("=> Start to inference")
    hits1 = hits5 = n = 0
    keep = []

    def serial():
        for i in range(0, len(items), batch_size):
            chunk = items[i:i + batch_size]
            x = torch.stack([preprocess(Image.open(p), size, mean, std) for p, _ in chunk]).to(device)
            yield x, torch.tensor([c for _, c in chunk], device=device), len(chunk)

    if processes is None:                  # default: decoding processes when the host has the cores for it and the run is long enough to
        # half of the usable cores (measured on a 16-core quota: 8 processes 3.3-3.6 k images/s, 16 processes 2.6-2.8 k, 16 threads 2.9 k)
        cores = effective_cores()
        processes = min(32, cores // 2) if (workers is None and len(items) >= 2048 and cores >= 8) else 0
    pipe = ImagePipeline(items, batch_size, device, size, mean, std, workers, processes=processes) if (workers != 0 or processes) else None
    t_start = time.perf_counter()
    for x, y, nb in (pipe if pipe is not None else serial()):
        chunk = range(nb)
        logits = 100.0 * model.encode_image(x).float() @ W
        if return_logits:
            keep.append(logits.float().cpu())
        a1, a5 = accuracy(logits, y, (1, min(5, logits.shape[1])))
        hits1 += a1 * len(chunk) / 100.0
        hits5 += a5 * len(chunk) / 100.0
        n += len(chunk)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if pipe is not None:
        pipe.close()
    top1, top5 = 100.0 * hits1 / max(n, 1), 100.0 * hits5 / max(n, 1)
    log("=> {:.1f} images/s end to end ({} images, {})".format(n / max(elapsed, 1e-9), n,
        (f"{pipe.processes} decoding processes" if pipe.processes else f"{pipe.workers} decoding threads") + ", pinned staging, copy stream"
        if pipe is not None else "single-threaded loader"))
    log("=> {dataset}% TEST:\tError@1 {error1:.3f}%\t{metric}@1 {top1:.3f}%\t".format(
        dataset=dataset, metric=metric, top1=top1, error1=100.0 - top1) + "accuracy@5 {:.3f}%\t({} images)".format(top5, n))
    res = dict(top1=top1, top5=top5, n=n, images_per_s=n / max(elapsed, 1e-9), elapsed_s=elapsed,
               loader_stats=dict(pipe.stats) if pipe is not None else None, loader_threads=(pipe.workers if pipe.processes == 0 else 0) if pipe is not None else 0,
               loader_processes=pipe.processes if pipe is not None else 0)
    if return_logits:
        res["logits"], res["classifier"] = torch.cat(keep), W.float().cpu()
        res["labels"] = torch.tensor([c for _, c in items])
    return res
