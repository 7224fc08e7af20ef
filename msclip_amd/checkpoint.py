"""Checkpoint I/O for the reference's file formats.

Eval loads a bare state_dict strictly (tools/zero_shot.py:223-224); the trainer helpers write either a bare state_dict
(`save_model_on_master`, lib/utils/utils.py:203-215) or a dict with a 'state_dict' entry (`save_checkpoint_on_master`,
:157-200), possibly with DDP's 'module.' prefix.  Shared tensors are serialised under both tower names; after loading
the aliases must still be one storage and hold identical values.
"""
import torch


def extract_state_dict(obj):
    sd = obj.get("state_dict", obj) if isinstance(obj, dict) else obj
    if not isinstance(sd, dict):
        raise TypeError("checkpoint holds no state_dict")
    if sd and all(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    return sd


def check_aliases(model, sd=None, atol=0.0):
    """The text-tower copies of the shared tensors must alias the visual ones (and agree in the file)."""
    vis, txt = model.visual.transformer.resblocks, model.transformer.resblocks
    start = model.share_from_layer
    if start is None:
        return 0
    n = 0
    for i in range(max(start, 1), len(vis)):
        for leaf in ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                     "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias"):
            a = vis[i].get_parameter(leaf)
            b = txt[i].get_parameter(leaf)
            if a.data_ptr() != b.data_ptr():
                raise RuntimeError(f"layer {i} {leaf}: text and visual tensors are not aliased")
            if sd is not None:
                fa, fb = sd[f"visual.transformer.resblocks.{i}.{leaf}"], sd[f"transformer.resblocks.{i}.{leaf}"]
                if (fa.float() - fb.float()).abs().max().item() > atol:
                    raise RuntimeError(f"checkpoint disagrees with itself on the shared tensor {leaf} of layer {i}")
            n += 1
    return n


def load_pretrained(model, path, strict=True, map_location="cpu"):
    sd = extract_state_dict(torch.load(path, map_location=map_location, weights_only=False))
    result = model.load_state_dict(sd, strict=strict)
    check_aliases(model, sd)
    return result


def save_model(model, path):
    """Bare state_dict, fp32, CPU -- what the reference's eval expects."""
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)
