"""Zero-shot evaluation CLI: counterpart of the reference's tools/eval_zeroshot.py (:14-31, the dataset loop) and
tools/zero_shot.py (:183-310, one dataset), with the same command line:

    python tools/eval_zeroshot.py --model experiments/model/b32-yfcc-msclips.yaml [--ds imagenet | --ds path/to/ds.yaml]
                                  [KEY VALUE ...]            # e.g. DATASET.ROOT /data/imagenet/ TEST.BATCH_SIZE_PER_GPU 64

`--ds` is a comma-separated list of dataset names (resolved to experiments/dataset/<name>.yaml) or yaml paths; without
it every known dataset runs (eval_zeroshot.py:43-46).  Per dataset the config is built like zero_shot.py:185-190:
update_config(dataset yaml), then update_config(model yaml), NAME cleared.  Images come from
DATASET.ROOT/DATASET.TEST_SET in ImageFolder layout; the checkpoint from MODEL.PRETRAINED_MODEL (strict load).

Extras (not in the reference): --max-images N and --max-classes C take a subset (BASELINE config C1's 64-image
plumbing run: the full 1000 x 80 prompt classifier is 80 000 text forwards), --ckpt / --bpe override the yaml /
packaged data, --random-init evaluates without a checkpoint (plumbing only).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msclip_amd import checkpoint, zeroshot                       # noqa: E402
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model      # noqa: E402
from msclip_amd.config import default_config, update_config      # noqa: E402
from msclip_amd.tokenizer import SimpleTokenizer                  # noqa: E402

cfg_files_dataset = {"imagenet": "experiments/dataset/imagenet.yaml"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Zeroshot Eval")
    ap.add_argument("--ds", type=str, default=None, help="Evaluation dataset configure file name(s).")
    ap.add_argument("--model", required=True, type=str, help="Evaluation model configure file name")
    ap.add_argument("--save-feature", default=False, type=str, help="accepted for command-line compatibility; unused")
    ap.add_argument("--ckpt", default=None, help="defaults to MODEL.PRETRAINED_MODEL of the yaml")
    ap.add_argument("--random-init", action="store_true", help="no checkpoint: plumbing run on random-init weights")
    ap.add_argument("--bpe", default=None, help="merges table (default: the packaged one)")
    ap.add_argument("--max-images", type=int, default=None)
    ap.add_argument("--max-classes", type=int, default=None, help="use only the first C class directories / names")
    ap.add_argument("--workers", type=int, default=None,
                    help="decoding threads of the input pipeline (default: config WORKERS, the reference's DataLoader setting, "
                         "zero_shot.py:70-81; 0 = the single-threaded loader)")
    ap.add_argument("--processes", type=int, default=None,
                    help="decoding PROCESSES instead of threads (default: min(32, cores / 2) for runs of >= 2048 images; 0 = threads)")
    ap.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)
    return ap.parse_args(argv)


def resolve_dataset(name):
    if os.path.exists(name):
        return name
    rel = cfg_files_dataset.get(name)
    for cand in ([rel, os.path.join(ROOT, rel)] if rel else []):
        if os.path.exists(cand):
            return cand
    raise Exception(f"Dataset {name} does not exist.")                     # eval_zeroshot.py:49-50


def build_config(ds_yaml, model_yaml, opts):
    """zero_shot.py:183-190."""
    class _A:
        pass
    a = _A()
    config = default_config()
    a.cfg, a.opts = ds_yaml, opts
    update_config(config, a)
    a.cfg = model_yaml
    update_config(config, a)
    config.defrost()
    config.NAME = ""
    config.freeze()
    return config


def zero_shot(args, ds_yaml, log=print):
    config = build_config(ds_yaml, args.model, args.opts or [])
    model = get_clip_model(config)
    model_file = args.ckpt or config.MODEL.PRETRAINED_MODEL
    if args.random_init:
        log("=> WARNING: --random-init, evaluating untrained weights (plumbing run)")
    else:
        log("=> load model file: {}".format(model_file))
        checkpoint.load_pretrained(model, model_file)                     # strict 521-key load (zero_shot.py:222-224)
    model = model.cuda().eval()
    log("=> switch to eval mode")
    classes, templates = zeroshot.load_prompts(zeroshot.prompt_name(config.DATASET.DATASET))
    val_root = os.path.join(config.DATASET.ROOT, config.DATASET.TEST_SET)
    log("=> Start to build zeroshot classifier")
    res = zeroshot.evaluate(model, SimpleTokenizer(args.bpe), val_root, classes, templates,
                            batch_size=config.TEST.BATCH_SIZE_PER_GPU, max_images=args.max_images,
                            max_classes=args.max_classes, size=config.TEST.IMAGE_SIZE[0], mean=config.INPUT.MEAN,
                            std=config.INPUT.STD, dataset=config.DATASET.DATASET,
                            metric=config.TEST.get("METRIC", "accuracy"), log=log,
                            workers=args.workers if args.workers is not None else None, processes=args.processes)
    return res


def run_jobs(argv=None):
    args = parse_args(argv)
    datasets = list(cfg_files_dataset.keys()) if args.ds is None else args.ds.split(",")
    files = [resolve_dataset(d) for d in datasets]                        # check availability first (eval_zeroshot.py:47-50)
    return [zero_shot(args, f) for f in files]


if __name__ == "__main__":
    run_jobs()
