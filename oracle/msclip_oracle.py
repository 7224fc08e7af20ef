"""CPU oracle for the MS-CLIP-S hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product (msclip_amd/) never does, and fails loudly when the HIP
extension is missing instead of falling back to anything in here.

This is a plain fp32 torch restatement of the algorithm of the reference's
released MS-CLIP-S path, written functionally (batch-first [B, L, C] tokens,
no nn.Module, weights addressed by their reference state_dict key).  Each
function cites the reference lines it follows; `M.py` abbreviates
/root/reference/lib/models/clip_openai_pe_res_v1.py.

PINNING: tools/make_golden.py imports the real reference in the build
container (the only place /root/reference exists), loads deterministic
synthetic weights into it, and stores its outputs under tests/golden/;
tests/test_oracle_golden.py checks this oracle against those vectors
(fp32, <=2e-5 abs on unit-norm features) and tests/test_oracle_vs_reference.py
re-checks against the live import when the reference tree is present.
The symmetric cross-entropy (contrastive_loss) has NO reference implementation
(SURVEY.md s8 a14): parity unpinned, pinned only against
torch.nn.functional.cross_entropy.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass
class Arch:
    """Hyper-parameters the released yamls select (experiments/model/*.yaml)."""
    embed_dim: int = 512
    image_size: int = 224
    patch_size: int = 32
    width: int = 768
    vision_layers: int = 12           # slot 0 is the conv stem (M.py:2040-2051)
    text_layers: int = 12
    heads: int = 12
    context_length: int = 77
    vocab_size: int = 49408
    stem_strides: Sequence[int] = (2, 2, 2, 2)          # EARLY_CONV_RES_STRIDES  (M.py:2277)
    parallel_strides: Sequence[int] = (2, 2, 2, 2, 2)   # PARALLEL_STRIDES        (M.py:2135)
    lateral_layers: Sequence[int] = (2, 4, 6, 8, 10)    # PARALLEL_LATERAL_LAYER
    t2b_kernels: Sequence[int] = (16, 8, 4, 2, 1)       # PRALLEL_T2B_KERNELS
    t2b_strides: Sequence[int] = (16, 8, 4, 2, 1)       # PRALLEL_T2B_STRIDES
    t2b_paddings: Sequence[int] = (0, 0, 0, 0, 0)
    t2b_usecls: bool = True                              # PRALLEL_T2B_USECLS
    share_from_layer: int = 1                            # N_LAYERS
    patch_conv: bool = False                             # EARLY_CONV off: plain patch convolution `visual.conv1` (M.py:2502-2508)

    @property
    def grid(self):
        return self.image_size // self.patch_size

    @property
    def image_tokens(self):
        return self.grid * self.grid + 1


def arch_b32():
    return Arch()


def arch_l16():
    """experiments/model/l16-fp8-msclips.yaml of this build (BASELINE config C5's stand-in; the reference builds it from the
    same yaml, tests/golden/l16-fp8-msclips.npz): ViT-L width / depth / heads on the 14 x 14 grid of the B/16 configs."""
    a = arch_b16()
    a.embed_dim, a.width, a.heads, a.vision_layers, a.text_layers = 768, 1024, 16, 24, 24
    return a


def arch_l14():
    """experiments/model/l14-fp8-msclips.yaml (BASELINE config C5): ViT-L/14 with the reference's plain patch convolution
    (M.py:2502-2508, VisualTransformer.forward :2655-2668), 16 x 16 grid, no parallel branch / adapters; every vision slot is an
    attention block."""
    return Arch(embed_dim=768, patch_size=14, width=1024, heads=16, vision_layers=24, text_layers=24, lateral_layers=(),
                patch_conv=True)


def arch_b16():
    return Arch(patch_size=16, stem_strides=(2, 2, 2, 1), parallel_strides=(2, 2, 2, 2, 1),
                t2b_kernels=(8, 4, 2, 1, 1), t2b_strides=(8, 4, 2, 1, 1))


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------

def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-12) -> Tensor:
    """TF-style LN, eps inside the sqrt, biased variance, fp32 math (M.py:204-219)."""
    xf = x.float()
    u = xf.mean(-1, keepdim=True)
    s = (xf - u).pow(2).mean(-1, keepdim=True)
    return w * ((xf - u) / torch.sqrt(s + eps)).to(x.dtype) + b


def quick_gelu(x: Tensor) -> Tensor:
    """x * sigmoid(1.702 x) (M.py:222-224)."""
    return x * torch.sigmoid(1.702 * x)


def batch_norm(x: Tensor, sd: SD, prefix: str, eps: float) -> Tensor:
    """Eval-mode BatchNorm2d from running stats on an NCHW tensor."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    mu, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    scale = w / torch.sqrt(var + eps)
    return x * scale[None, :, None, None] + (b - mu * scale)[None, :, None, None]


def causal_mask(n: int, device=None) -> Tensor:
    """Additive -inf strictly-upper-triangular mask (M.py:2965-2971)."""
    return torch.full((n, n), float("-inf"), device=device).triu_(1)


def attention(x: Tensor, sd: SD, p: str, heads: int, mask: Tensor = None) -> Tensor:
    """Self-attention of Attention_CUST (M.py:592-612, 705-755), batch-first.

    q is scaled by head_dim**-0.5 BEFORE q.k^T (M.py:707); softmax in the input
    dtype; additive mask; dropout p=0."""
    B, L, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = q * (float(hd) ** -0.5)
    q = q.reshape(B, L, heads, hd).transpose(1, 2)
    k = k.reshape(B, L, heads, hd).transpose(1, 2)
    v = v.reshape(B, L, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    o = torch.softmax(s, dim=-1) @ v
    o = o.transpose(1, 2).reshape(B, L, C)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def mlp(x: Tensor, sd: SD, p: str) -> Tensor:
    """c_fc -> QuickGELU -> c_proj (M.py:794-798)."""
    h = F.linear(x, sd[p + ".c_fc.weight"], sd[p + ".c_fc.bias"])
    return F.linear(quick_gelu(h), sd[p + ".c_proj.weight"], sd[p + ".c_proj.bias"])


def residual_block(x: Tensor, sd: SD, p: str, heads: int, mask: Tensor = None) -> Tensor:
    """x += attn(ln_1 x); x += mlp(ln_2 x); drop_path is Identity (M.py:1027-1028)."""
    x = x + attention(layer_norm(x, sd[p + ".ln_1.weight"], sd[p + ".ln_1.bias"]), sd, p + ".attn", heads, mask)
    x = x + mlp(layer_norm(x, sd[p + ".ln_2.weight"], sd[p + ".ln_2.bias"]), sd, p + ".mlp")
    return x


# ----------------------------------------------------------------------------
# vision tower
# ----------------------------------------------------------------------------

def stem(img: Tensor, sd: SD, arch: Arch, taps: dict = None) -> Tensor:
    """EarlyconvRes (M.py:1939-2000) with ResBasicBlock_v0 stages (M.py:1898-1936).

    conv3x3 s2 3->w/16, BN(1e-5), ReLU; 4x relu(BN(conv3x3 s) + BN(conv1x1 s));
    1x1 last_conv without BN/ReLU.  NCHW in, [B, width, g, g] out."""
    p = "visual.transformer.resblocks.0"
    x = F.conv2d(img, sd[p + ".conv1.weight"], stride=2, padding=1)
    x = F.relu(batch_norm(x, sd, p + ".bn1", 1e-5))
    if taps is not None:
        taps["stem_conv1"] = x
    for i, s in enumerate(arch.stem_strides):
        q = f"{p}.resnet_stage.conv_{i}"
        main = batch_norm(F.conv2d(x, sd[q + ".conv1.weight"], stride=s, padding=1), sd, q + ".bn1", 1e-5)
        short = batch_norm(F.conv2d(x, sd[q + ".downsample.0.weight"], stride=s), sd, q + ".downsample.1", 1e-5)
        x = F.relu(main + short)
        if taps is not None:
            taps[f"stem_stage{i}"] = x
    return F.conv2d(x, sd[p + ".last_conv.weight"])


def parallel_stage(x: Tensor, sd: SD, arch: Arch, j: int) -> Tensor:
    """Stage j of the parallel conv branch.

    j == 0: conv3x3 s2 + BN(1e-5) + ReLU on the raw image (M.py:2260-2273).
    j >= 1: one bottleneck ConvResBlock, BN eps 1e-6, mid = out/2, projection
    shortcut with the stage stride (M.py:1812-1861, 1864-1895)."""
    p = f"visual.transformer.parallel_branch_v.{j}"
    s = arch.parallel_strides[j]
    if j == 0:
        x = F.conv2d(x, sd[p + ".conv.weight"], stride=s, padding=1)
        return F.relu(batch_norm(x, sd, p + ".bn", 1e-5))
    q = p + ".resnet_stage.conv_0"
    y = F.relu(batch_norm(F.conv2d(x, sd[q + ".conv1.weight"]), sd, q + ".bn1", 1e-6))
    y = F.relu(batch_norm(F.conv2d(y, sd[q + ".conv2.weight"], stride=s, padding=1), sd, q + ".bn2", 1e-6))
    y = batch_norm(F.conv2d(y, sd[q + ".conv3.weight"]), sd, q + ".bn3", 1e-6)
    r = batch_norm(F.conv2d(x, sd[q + ".residual_conv.weight"], stride=s), sd, q + ".residual_bn", 1e-6)
    return F.relu(y + r)


def lateral_adapter(top: Tensor, x: Tensor, sd: SD, arch: Arch, j: int) -> Tensor:
    """Lateral_Adapter.forward (M.py:1752-1778), batch-first tokens.

    t = pw1x1(BN(dwconv_{k=s}(top))) -> tokens; grid tokens -> BN(dw3x3) ;
    the cls token is concatenated to BOTH operands (USECLS), so row 0 of the
    sum is 2*cls (M.py:1766-1771); out = ln_adapt(sum)."""
    p = f"visual.transformer.parallel_lateral_adapter.{j}"
    B, L, C = x.shape
    g = arch.grid
    k, s, pad = arch.t2b_kernels[j], arch.t2b_strides[j], arch.t2b_paddings[j]
    t = F.conv2d(top, sd[p + ".top2bottom_dw_conv.conv.weight"], stride=s, padding=pad, groups=top.shape[1])
    t = batch_norm(t, sd, p + ".top2bottom_dw_conv.bn", 1e-5)
    t = F.conv2d(t, sd[p + ".top2bottom_pw_conv.conv.weight"])
    assert t.shape[-2:] == (g, g), (t.shape, g, k)
    t = t.flatten(2).transpose(1, 2)                                   # b (h w) c
    cls, grid = x[:, :1], x[:, 1:]
    grid = grid.transpose(1, 2).reshape(B, C, g, g)
    bo = F.conv2d(grid, sd[p + ".bottom_dw_conv.conv.weight"], padding=1, groups=C)
    bo = batch_norm(bo, sd, p + ".bottom_dw_conv.bn", 1e-5).flatten(2).transpose(1, 2)
    bo = torch.cat([cls, bo], 1)
    t = torch.cat([cls if arch.t2b_usecls else torch.zeros_like(cls), t], 1)
    return layer_norm(bo + t, sd[p + ".ln_adapt.weight"], sd[p + ".ln_adapt.bias"])


def image_tokens(img: Tensor, sd: SD, arch: Arch, taps: dict = None) -> Tensor:
    """Slot 0 of the visual Transformer: stem -> tokens, cls, +pos, ln_pre
    (M.py:2416-2426)."""
    if arch.patch_conv:                                   # M.py:2657: conv1 with kernel == stride == patch, no bias
        x = F.conv2d(img, sd["visual.conv1.weight"], stride=arch.patch_size)
    else:
        x = stem(img, sd, arch, taps)
    B = x.shape[0]
    x = x.flatten(2).transpose(1, 2)
    cls = sd["visual.class_embedding"].to(x.dtype).expand(B, 1, -1)
    x = torch.cat([cls, x], 1) + sd["visual.positional_embedding"]
    return layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])


def encode_image(img: Tensor, sd: SD, arch: Arch, norm: bool = True, taps: dict = None, block_fn=None) -> Tensor:
    """CLIP.encode_image (M.py:2979-2985) -> VisualTransformer.forward
    (M.py:2621-2638, 2669-2697) -> Transformer.forward (M.py:2388-2459).
    block_fn: stand-in for residual_block (oracle/fp8_recipe.py emulates PRECISION fp8 through it)."""
    residual_block = block_fn or globals()["residual_block"]
    img = img.float()
    x = image_tokens(img, sd, arch, taps)
    if taps is not None:
        taps["tokens_ln_pre"] = x
    par = img
    for idx in range(0 if arch.patch_conv else 1, arch.vision_layers):      # slot 0 is the conv stem unless the patch conv tokenises
        if idx in arch.lateral_layers:
            j = list(arch.lateral_layers).index(idx)
            par = parallel_stage(par, sd, arch, j)
            x = lateral_adapter(par, x, sd, arch, j)
            if taps is not None:
                taps[f"parallel{j}"] = par
                taps[f"adapter{j}"] = x
        x = residual_block(x, sd, f"visual.transformer.resblocks.{idx}", arch.heads)
        if taps is not None:
            taps[f"vblock{idx}"] = x
    x = layer_norm(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    x = x @ sd["visual.proj"]
    return x / x.norm(dim=-1, keepdim=True) if norm else x


# ----------------------------------------------------------------------------
# text tower
# ----------------------------------------------------------------------------

def encode_text(text: Tensor, sd: SD, arch: Arch, norm: bool = True, taps: dict = None, block_fn=None) -> Tensor:
    """CLIP.encode_text (M.py:3043-3079): embed + pos, 12 causal blocks, row at
    argmax(token id) (EOT has the largest id), ln_final, @text_projection, L2."""
    residual_block = block_fn or globals()["residual_block"]
    x = sd["token_embedding.weight"][text] + sd["positional_embedding"]
    mask = causal_mask(text.shape[1], text.device)
    for i in range(arch.text_layers):
        x = residual_block(x, sd, f"transformer.resblocks.{i}", arch.heads, mask)
        if taps is not None:
            taps[f"tblock{i}"] = x
    x = x[torch.arange(x.shape[0]), text.argmax(dim=-1)]
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    x = x @ sd["text_projection"]
    return x / x.norm(dim=-1, keepdim=True) if norm else x


# ----------------------------------------------------------------------------
# contrastive head
# ----------------------------------------------------------------------------

def clip_logits(f_img_all: Tensor, f_txt_all: Tensor, logit_scale: Tensor) -> Tensor:
    """logits = exp(logit_scale) * I_all @ T_all^T (M.py:3136-3141)."""
    return logit_scale.exp() * f_img_all @ f_txt_all.t()


def forward(img: Tensor, text: Tensor, sd: SD, arch: Arch) -> Tensor:
    """CLIP.forward at world size 1 (M.py:3126-3155)."""
    return clip_logits(encode_image(img, sd, arch), encode_text(text, sd, arch), sd["logit_scale"])


def gather_rank_major(per_rank: List[Tensor]) -> Tensor:
    """What gather_tensors returns on every rank: rank-major concatenation
    (lib/utils/comm.py:140-154)."""
    return torch.cat(per_rank, dim=0)


def contrastive_loss(logits: Tensor) -> Tensor:
    """Standard CLIP symmetric CE over the global batch.  NOT IN THE REFERENCE
    (parity unpinned): 0.5*(CE(logits, arange) + CE(logits^T, arange))."""
    n = logits.shape[0]
    lab = torch.arange(n, device=logits.device)
    return 0.5 * (F.cross_entropy(logits, lab) + F.cross_entropy(logits.t(), lab))


def zeroshot_classifier(prompt_tokens_per_class: List[Tensor], sd: SD, arch: Arch) -> Tensor:
    """tools/zero_shot.py:122-134: per class mean of unit text features, renormalised,
    stacked as columns -> W[embed, n_classes]."""
    cols = []
    for toks in prompt_tokens_per_class:
        e = encode_text(toks, sd, arch)
        e = e.mean(dim=0)
        cols.append(e / e.norm())
    return torch.stack(cols, dim=1)
