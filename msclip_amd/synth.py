"""Deterministic synthetic weights and inputs.

There is no checkpoint and no dataset offline, so tests,
fixtures and the benchmark all use weights produced by this generator.  A value
depends only on (seed, canonical state_dict key, shape): the build container
fills the imported reference with it to make tests/golden/, and the GPU box
fills the product model with the very same numbers.

Shared tensors are serialised under both tower names in the reference's
state_dict; the text-side alias is mapped to the visual name
before hashing so both names get identical values.
"""
import re
import zlib

import numpy as np
import torch

_SHARED_LEAF = re.compile(
    r"^transformer\.resblocks\.(\d+)\.(attn\.in_proj_weight|attn\.in_proj_bias|attn\.out_proj\.(?:weight|bias)"
    r"|mlp\.c_fc\.(?:weight|bias)|mlp\.c_proj\.(?:weight|bias))$")


def canonical_key(key, share_from_layer=1, vision_layers=12):
    m = _SHARED_LEAF.match(key)
    if m and share_from_layer <= int(m.group(1)) < vision_layers:
        return "visual." + key
    return key


def _rng(seed, key):
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def _is_bn(key):
    parts = key.split(".")
    owner = parts[-2]
    return owner.startswith("bn") or owner == "residual_bn" or (owner == "1" and parts[-3] == "downsample")


def synth_tensor(key, shape, dtype=torch.float32, seed=0, share_from_layer=1, vision_layers=12):
    key = canonical_key(key, share_from_layer, vision_layers)
    r = _rng(seed, key)
    leaf = key.split(".")[-1]
    shape = tuple(shape)
    if leaf == "num_batches_tracked":
        return torch.tensor(1000, dtype=torch.int64)
    if leaf == "running_mean":
        a = r.normal(0.0, 0.1, shape)
    elif leaf == "running_var":
        a = r.uniform(0.5, 1.5, shape)
    elif key == "logit_scale":
        a = np.full(shape, np.log(1.0 / 0.07))
    elif leaf in ("class_embedding", "proj", "text_projection") or key == "visual.positional_embedding":
        width = shape[0] if leaf in ("proj", "text_projection") else shape[-1]
        a = r.standard_normal(shape) * width ** -0.5
    elif key == "positional_embedding":
        a = r.standard_normal(shape) * 0.01
    elif key == "token_embedding.weight":
        a = r.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
    elif len(shape) == 4:                                     # conv: kaiming so ReLU stacks keep O(1) activations
        fan_in = shape[1] * shape[2] * shape[3]
        a = r.standard_normal(shape) * np.sqrt(2.0 / fan_in)
    elif len(shape) == 1 and leaf == "weight":                # LayerNorm / BatchNorm gamma
        a = r.uniform(0.8, 1.2, shape)
    elif len(shape) == 1 and leaf == "bias" and (_is_bn(key) or ".ln_" in key or key.startswith("ln_")):
        a = r.normal(0.0, 0.1, shape)
    elif leaf == "in_proj_weight":
        a = r.standard_normal(shape) * 0.05
    elif leaf in ("in_proj_bias", "bias"):
        a = r.normal(0.0, 0.05, shape)
    elif len(shape) == 2:                                     # out_proj / c_fc / c_proj
        a = r.standard_normal(shape) * (0.03 if ".c_fc." in key else 0.02)
    else:
        raise KeyError(f"no synthetic rule for {key} {shape}")
    return torch.from_numpy(np.ascontiguousarray(a)).reshape(shape).to(dtype)


def synth_state_dict(schema, seed=0, share_from_layer=1):
    """schema: iterable of (key, shape, dtype) in state_dict order."""
    out = {}
    schema = list(schema)
    vl = re.compile(r"^visual\.transformer\.resblocks\.(\d+)\.")
    vision_layers = 1 + max([int(m.group(1)) for m in (vl.match(k) for k, _, _ in schema) if m] or [11])
    for key, shape, dtype in schema:
        out[key] = synth_tensor(key, shape, dtype, seed, share_from_layer, vision_layers)
    return out


def schema_of(module_or_sd):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return [(k, tuple(v.shape), v.dtype) for k, v in sd.items()]


def synth_images(batch, size=224, seed=0):
    """N(0,1) pixels ~ ImageNet-normalised images."""
    r = np.random.default_rng([seed, 0x1A6E])
    return torch.from_numpy(r.standard_normal((batch, 3, size, size), dtype=np.float32))


def synth_tokens(batch, context_length=77, vocab_size=49408, seed=1, min_len=4, max_len=60):
    """SOT, U{min..max} random ids, EOT (= vocab-1, the largest id so argmax finds it), zero pad."""
    r = np.random.default_rng([seed, 0x70C5])
    sot, eot = vocab_size - 2, vocab_size - 1
    tok = np.zeros((batch, context_length), dtype=np.int64)
    max_len = min(max_len, context_length - 2)
    for b in range(batch):
        n = int(r.integers(min_len, max_len + 1))
        tok[b, 0] = sot
        tok[b, 1:1 + n] = r.integers(1, sot, n)
        tok[b, 1 + n] = eot
    return torch.from_numpy(tok)
