"""Worker process of msclip_amd.zeroshot.ImagePipeline (processes > 0): decodes image files into a shared-memory staging area.

    python -m msclip_amd._decode_worker <shm name> <depth> <batch> <size>

Protocol: one JSON task per stdin line, [batch index, staging slot, first row, [paths]]; one JSON answer per task on stdout,
[batch index, number of files, error text or null].  Imports numpy + PIL only -- no torch, no HIP: the process starts in ~0.2 s,
never touches the GPU, and is a plain subprocess (neither a fork of a parent with a live HIP runtime nor a multiprocessing
child that would re-import the parent's __main__)."""
import json
import sys

import numpy as np


def load_pixels(path, size=224):
    """Image file -> uint8 [size, size, 3]: decode, shorter side to `size` (bicubic), centre crop -- the geometric half of the
    reference's transform chain (tools/zero_shot.py:202-207: Resize(size, bicubic), CenterCrop(size))."""
    from PIL import Image
    with Image.open(path) as im:
        img = im.convert("RGB")
    w, h = img.size
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    img = img.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    return np.array(img.crop((left, top, left + size, top + size)), dtype=np.uint8)


def main(argv):
    from multiprocessing import resource_tracker, shared_memory
    name, depth, batch, size = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    shm = shared_memory.SharedMemory(name=name)
    try:
        resource_tracker.unregister(shm._name, "shared_memory")     # (python < 3.13 would unlink the PARENT's segment at our exit)
    except Exception:
        pass
    buf = np.ndarray((depth, batch, size, size, 3), dtype=np.uint8, buffer=shm.buf)
    out = sys.stdout
    try:
        for line in sys.stdin:
            line = line.strip()
            if not line:
                continue
            b, k, j0, paths = json.loads(line)
            err = None
            try:
                for j, p in enumerate(paths):
                    buf[k, j0 + j] = load_pixels(p, size)
            except Exception as exc:                        # the parent re-raises it with the files' names
                err = f"{type(exc).__name__}: {exc} ({paths})"
            out.write(json.dumps([b, len(paths), err]) + "\n")
            out.flush()
    finally:
        del buf
        shm.close()


if __name__ == "__main__":
    main(sys.argv[1:])
