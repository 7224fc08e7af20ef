"""Config surface of the reference, without yacs (not installed here).

Mirrors what the hot path needs from lib/config/default.py: a CfgNode with
attribute access, `merge_from_file` / `merge_from_list`, `defrost` / `freeze`,
`BASE:` inheritance resolved relative to the yaml (default.py:279-291) and
`update_config(config, args)` (default.py:294-319).  Only the keys the model
factory reads are given defaults; MODEL, MODEL.SPEC, CUSTOM,
TEST and DATASET accept new keys like the reference's `new_allowed` nodes.
"""
import copy
import os

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self._frozen:
            raise AttributeError(f"config is frozen, cannot set {k}")
        self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def _each(self):
        for v in self.values():
            if isinstance(v, CfgNode):
                yield v

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for n in self._each():
            n.defrost()

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for n in self._each():
            n.freeze()

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        n = CfgNode()
        for k, v in self.items():
            n[k] = copy.deepcopy(v, memo)
        return n

    def merge_from_other(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_other(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_other(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, CfgNode())
            node[parts[-1]] = yaml.safe_load(val) if isinstance(val, str) else val


def default_config():
    c = CfgNode()
    c.BASE = [""]
    c.NAME = ""
    c.OUTPUT_DIR = "OUTPUT/"
    c.RANK = 0
    c.DIST_BACKEND = "nccl"            # == RCCL on ROCm
    c.WORKERS = 4                      # decoding threads of the eval input pipeline (default.py:28; the zero-shot DataLoader uses 6)
    c.MODEL = CfgNode({"NAME": "clip_openai_pe_res_v1", "PRETRAINED_MODEL": "", "SPEC": {}})
    # trainer keys the optimizer set-up of the (unreleased) trainer reads (default.py:126-133, 189-190)
    c.TRAIN = CfgNode({"IMAGE_SIZE": [224, 224], "BATCH_SIZE_PER_GPU": 256, "LR": 1e-3, "SCALE_LR": True,
                       "OPTIMIZER": "sgd", "OPTIMIZER_ARGS": {}, "MOMENTUM": 0.9, "WD": 1e-4, "WITHOUT_WD_LIST": [],
                       "LR_SCHEDULER": {}})
    c.TEST = CfgNode({"IMAGE_SIZE": [224, 224], "BATCH_SIZE_PER_GPU": 32, "MODEL_FILE": "", "CENTER_CROP": True,
                      "INTERPOLATION": 3})
    c.INPUT = CfgNode({"MEAN": [0.485, 0.456, 0.406], "STD": [0.229, 0.224, 0.225]})  # default.py:84-85
    c.DATASET = CfgNode({"DATASET": "imagenet", "ROOT": "", "TEST_SET": "val"})
    c.CUSTOM = CfgNode({"LR_SHARE": 0.0, "WD_SHARE": 0.0})
    return c


def _merge_with_base(cfg, path):
    """BASE entries are yaml paths relative to the including file and are merged first."""
    with open(path) as f:
        top = yaml.safe_load(f) or {}
    for base in top.get("BASE", []) or []:
        if base:
            _merge_with_base(cfg, os.path.join(os.path.dirname(path), base))
    top.pop("BASE", None)
    cfg.merge_from_other(top)


def update_config(config, args):
    """`args` needs `.cfg` (yaml path) and optionally `.opts` (KEY VALUE list)."""
    config.defrost()
    _merge_with_base(config, args.cfg)
    opts = getattr(args, "opts", None)
    if opts:
        config.merge_from_list(opts)
    if config.TRAIN.get("SCALE_LR", False):                 # default.py:299-304: linear LR scaling with the world size
        from .comm import comm
        config.TRAIN.LR = config.TRAIN.LR * comm.world_size
        if config.CUSTOM.get("LR_SHARE", False):
            config.CUSTOM.LR_SHARE = config.CUSTOM.LR_SHARE * comm.world_size
    name = os.path.splitext(os.path.basename(args.cfg))[0]
    config.NAME = name + config.NAME                        # default.py:305-306 (file name is a PREFIX)
    config.freeze()
    return config


def load_config(path, opts=None):
    class _A:
        pass
    a = _A()
    a.cfg, a.opts = path, opts
    return update_config(default_config(), a)


EXPERIMENTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments", "model")


def named_config(name, opts=None):
    """name: 'b32-yfcc-msclips', 'b32-laion-msclips', 'b16-yfcc-msclips'."""
    return load_config(os.path.join(EXPERIMENTS, name + ".yaml"), opts)
