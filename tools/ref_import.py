"""Import the upstream MS-CLIP model module from /root/reference (build container only).

TEST INFRASTRUCTURE.  This file never travels to the GPU box in any useful form:
/root/reference does not exist there.  It is used by tools/make_golden.py to
produce the committed fixtures under tests/golden/ and by the local-only
cross-check in tests/test_oracle_vs_reference.py (auto-skipped when the
reference tree is absent).

The upstream module needs four shims to import under torch 2.x (SURVEY.md
Appendix A): `transformers` must be imported before the timm stub exists,
`torch.nn.modules.linear._LinearWithBias` was removed in torch>=1.9, `timm`
is not installed (DropPath is identity at p=0 in every released config), and
`ftfy` is not installed (tokenizer only).
"""
import os
import sys
import types

import torch
import torch.nn as nn
import yaml

REF_ROOT = os.environ.get("MSCLIP_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "lib", "models", "clip_openai_pe_res_v1.py"))


class _Attr(dict):
    """Attribute-dict whose missing attributes raise AttributeError (the model
    reads every CUSTOM key through getattr(node, key, default))."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _wrap(d):
    if isinstance(d, dict):
        return _Attr({k: _wrap(v) for k, v in d.items()})
    return d


def _merge(a, b):
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            _merge(a[k], v)
        else:
            a[k] = v
    return a


def load_reference_config(name):
    """name: 'b32-yfcc-msclips' | 'b16-yfcc-msclips' | 'b32-laion-msclips'."""
    d = os.path.join(REF_ROOT, "experiments", "model")
    path = os.path.join(d, name + ".yaml")
    if not os.path.exists(path):       # a config of this build the reference can express (l16-fp8-msclips): its BASE is still the reference's
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments", "model", name + ".yaml")
    with open(path) as f:
        top = yaml.safe_load(f)
    cfg = {}
    for base in top.pop("BASE", []):
        with open(os.path.join(d, base)) as f:
            _merge(cfg, yaml.safe_load(f))
    _merge(cfg, top)
    return _wrap(cfg)


_MOD = None


def import_reference_module():
    global _MOD
    if _MOD is not None:
        return _MOD
    from transformers import AutoModel  # noqa: F401  (must precede the timm stub)
    import torch.nn.modules.linear as L
    if not hasattr(L, "_LinearWithBias"):
        L._LinearWithBias = L.NonDynamicallyQuantizableLinear

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    tl = types.ModuleType("timm.models.layers")
    tl.DropPath, tl.trunc_normal_ = DropPath, nn.init.trunc_normal_
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    sys.modules.setdefault("timm.models", types.ModuleType("timm.models"))
    sys.modules["timm.models.layers"] = tl
    if "ftfy" not in sys.modules:
        ftfy = types.ModuleType("ftfy")
        ftfy.fix_text = lambda s: s
        sys.modules["ftfy"] = ftfy
    lib = os.path.join(REF_ROOT, "lib")
    if lib not in sys.path:
        sys.path.insert(0, lib)
    import logging
    logging.disable(logging.INFO)  # the model logs one line per initialised module
    from models import clip_openai_pe_res_v1 as M
    _MOD = M
    return M


def build_reference_model(name):
    M = import_reference_module()
    cfg = load_reference_config(name)
    model = M.get_clip_model(cfg).eval()
    return model, cfg


def ensure_single_rank_group():
    """CLIP.forward all_gathers unconditionally when GATHER_TENSORS is True."""
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29591")
        dist.init_process_group("gloo", rank=0, world_size=1)
