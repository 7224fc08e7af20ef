"""Drop-in for the reference's `models.clip_openai_pe_res_v1` on the released
MS-CLIP-S path (reference lib/models/clip_openai_pe_res_v1.py, "M.py").

Same factory (`get_clip_model(config)`, M.py:3182), same module tree and
parameter names (so a reference checkpoint loads with strict=True, including
the text-tower aliases of the shared tensors, M.py:2786-2830), same methods
(`encode_image`, `encode_text`, `forward`, M.py:2979/3043/3126).  The modules
below only HOLD parameters; all arithmetic runs in the HIP kernels driven by
msclip_amd.engine.Engine.  There is no CPU execution path: calling the model
without a GPU (or without the built library) raises HipUnavailable.

Every experimental switch of the reference that no released config enables is
rejected with NotImplementedError instead of silently building another net.
"""
import logging

import torch
from torch import nn

from . import comm as _comm

_UNSUPPORTED_TRUTHY = (
    "LORA_OPEN", "GUMBEL_SELECT", "GUMBEL_ADDTWO", "SHARE_BOTTOM_LAYER", "SAVE_GRADIENT", "GET_GRADIENT_FROMCKPT",
    "LOAD_SEARCHED_ARCH", "CONVIT_IN_V", "CVT_IN_V", "ADAPTER_FLAG", "PERCEIVER_IN_V", "PERCEIVER_IN_T",
    "PARALLEL_REUSE_EARLYCONV_FIRSTLAYER", "PARALLEL_REUSE_EARLYCONV_ALLLAYER", "PARALLEL_B2T",
    "PARALLEL_T2B_WINDOWATTN", "PARALLEL_T2B_POOL_SIZE", "PRALLEL_T2B_ADD_BN_RELU", "PRALLEL_T2B_ADD_BN_LN_RELU",
    "PRALLEL_T2B_NOLN_ADD", "CONTAINER_IN_V", "OUTPUT_ATTN_RAW", "OUTPUT_BEFORE_ATTN", "OUTPUT_AFTER_ATTN",
    "OUTPUT_AFTER_ATTN_LN", "OUTPUT_LAST_LN", "LORA_INIT", "VISUAL_LAYER_MINUS1",
    # read by Attention_CUST through custom_config even without LORA_OPEN (M.py:346-395): LoRA adapters inside attention
    "LORA_ATTN_DIM", "VISUAL_LORA_LOCAL", "LORA_ALPHA", "LORA_DROPOUT", "LORA_R_DROPOUT",
)
# CUSTOM keys this build honours plus the trainer-only ones the model never reads.  Any OTHER key with a
# truthy value is an experimental switch of the reference (or a typo): rejected instead of silently building another net.
_ACCEPTED = {
    "CUSTOM_ATTN", "SHARE_MODULES", "N_LAYERS", "VISUAL_LAYER_MINUS1", "PARALLEL_IN_V", "PARALLEL_N_LAYERS",
    "PARALLEL_LATERAL_LAYER", "PARALLEL_KERNELS", "PARALLEL_PADDINGS", "PARALLEL_STRIDES", "PRALLEL_T2B_KERNELS",
    "PRALLEL_T2B_PADDINGS", "PRALLEL_T2B_STRIDES", "PRALLEL_T2B_USECLS", "PARALLEL_RESNET", "PARALLEL_RESNET_LAYERS",
    "EARLY_CONV", "EARLY_CONV_NEW_IMPLEMENT", "EARLY_CONV_RES", "EARLY_CONV_RES_FIRSTCONV_KERNEL", "EARLY_CONV_RES_BLOCK",
    "EARLY_CONV_RES_LAYERS", "EARLY_CONV_RES_STRIDES",
    "LR_SHARE", "WD_SHARE", "LORA_WHERE_ADD", "GUMBEL_LR", "WITHOUT_WD_LIST",      # trainer / inert defaults (default.py:189-191)
}


def _get(node, key, default=None):
    return getattr(node, key, default) if node is not None else default


def _ln(width):
    return nn.LayerNorm(width, eps=1e-12)          # parameter holder; TF-style eps (M.py:204-219)


class _Attn(nn.Module):
    """Parameter layout of Attention_CUST (M.py:253-296)."""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * d))
        self.out_proj = nn.Linear(d, d)


class ResidualAttentionBlock(nn.Module):
    """M.py:758-801: attn, ln_1, mlp(c_fc, gelu, c_proj), ln_2."""

    def __init__(self, d, heads):
        super().__init__()
        self.attn = _Attn(d)
        self.ln_1 = _ln(d)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(d, 4 * d))
        self.mlp.add_module("c_proj", nn.Linear(4 * d, d))
        self.ln_2 = _ln(d)
        self.heads = heads


class _ResBasicBlockV0(nn.Module):
    """M.py:1898-1936."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout))
        self.stride = stride


class EarlyconvRes(nn.Module):
    """M.py:1939-2000 (first_conv_k 3, 'basic_v0' blocks, one block per stage)."""

    def __init__(self, width, strides):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width // 16, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 16)
        self.resnet_stage = nn.Sequential()
        n = len(strides)
        for i, s in enumerate(strides):
            cin = width // (2 ** (n - i))
            self.resnet_stage.add_module(f"conv_{i}", _ResBasicBlockV0(cin, cin * 2, s))
        self.last_conv = nn.Conv2d(width, width, 1, bias=False)
        self.strides = list(strides)


class _ConvBnRelu(nn.Module):
    """M.py:2260-2273."""

    def __init__(self, cin, cout, k, pad, stride):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, pad, bias=False)
        self.bn = nn.BatchNorm2d(cout)


class _ConvResBlock(nn.Module):
    """M.py:1812-1861 with res_conv=True; BatchNorm eps 1e-6."""

    def __init__(self, cin, mid, cout, k, stride, pad):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid, eps=1e-6)
        self.conv2 = nn.Conv2d(mid, mid, k, stride, pad, bias=False)
        self.bn2 = nn.BatchNorm2d(mid, eps=1e-6)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout, eps=1e-6)
        self.residual_conv = nn.Conv2d(cin, cout, 1, stride, 0, bias=False)
        self.residual_bn = nn.BatchNorm2d(cout, eps=1e-6)


class _ResnetStage(nn.Module):
    """M.py:1864-1895 with num_layer == 1."""

    def __init__(self, cin, mid, cout, k, stride, pad):
        super().__init__()
        self.resnet_stage = nn.Sequential()
        self.resnet_stage.add_module("conv_0", _ConvResBlock(cin, mid, cout, k, stride, pad))


class LateralAdapter(nn.Module):
    """M.py:1539-1637 (top->bottom only, BN after the depthwise conv, ln_adapt)."""

    def __init__(self, top_dim, bottom_dim, k, pad, stride):
        super().__init__()
        self.top2bottom_dw_conv = nn.Sequential()
        self.top2bottom_dw_conv.add_module("conv", nn.Conv2d(top_dim, top_dim, k, stride, pad, bias=False, groups=top_dim))
        self.top2bottom_dw_conv.add_module("bn", nn.BatchNorm2d(top_dim))
        self.top2bottom_pw_conv = nn.Sequential()
        self.top2bottom_pw_conv.add_module("conv", nn.Conv2d(top_dim, bottom_dim, 1, bias=False))
        self.bottom_dw_conv = nn.Sequential()
        self.bottom_dw_conv.add_module("conv", nn.Conv2d(bottom_dim, bottom_dim, 3, 1, 1, bias=False, groups=bottom_dim))
        self.bottom_dw_conv.add_module("bn", nn.BatchNorm2d(bottom_dim))
        self.ln_adapt = _ln(bottom_dim)
        self.kernel, self.stride, self.padding = k, stride, pad


class Transformer(nn.Module):
    """M.py:2003-2258: `resblocks` (+ for the visual tower: conv stem in slot 0, `parallel_branch_v`,
    `parallel_lateral_adapter`)."""

    def __init__(self, width, layers, heads, custom, modality, first_conv=False):
        super().__init__()
        self.width, self.layers, self.modality, self.first_conv = width, layers, modality, first_conv
        blocks = []
        for i in range(layers):
            if first_conv and i == 0:
                if not _get(custom, "EARLY_CONV_RES", False):
                    raise NotImplementedError("EARLY_CONV without EARLY_CONV_RES is not part of the MS-CLIP-S path")
                if _get(custom, "EARLY_CONV_RES_FIRSTCONV_KERNEL", 3) != 3:
                    raise NotImplementedError("EARLY_CONV_RES_FIRSTCONV_KERNEL != 3")
                if _get(custom, "EARLY_CONV_RES_BLOCK", "basic_v0") != "basic_v0":
                    raise NotImplementedError("EARLY_CONV_RES_BLOCK != 'basic_v0'")
                if list(_get(custom, "EARLY_CONV_RES_LAYERS", [1, 1, 1, 1])) != [1, 1, 1, 1]:
                    raise NotImplementedError("EARLY_CONV_RES_LAYERS != [1, 1, 1, 1]")
                blocks.append(EarlyconvRes(width, _get(custom, "EARLY_CONV_RES_STRIDES", [2, 2, 2, 2])))
            else:
                blocks.append(ResidualAttentionBlock(width, heads))
        self.resblocks = nn.Sequential(*blocks)
        self.parallel_in_v = bool(_get(custom, "PARALLEL_IN_V", False)) and modality == "visual"
        if self.parallel_in_v:
            n = _get(custom, "PARALLEL_N_LAYERS", 0)
            self.parallel_lateral_layers = list(_get(custom, "PARALLEL_LATERAL_LAYER", []))
            if n != 5 or len(self.parallel_lateral_layers) != 5:
                raise NotImplementedError("MS-CLIP-S uses 5 parallel stages / lateral adapters")
            if not _get(custom, "PARALLEL_RESNET", False) or list(_get(custom, "PARALLEL_RESNET_LAYERS", [])) != [0, 1, 1, 1, 1]:
                raise NotImplementedError("PARALLEL_RESNET with PARALLEL_RESNET_LAYERS [0,1,1,1,1] is required")
            cin = [3, width // 16, width // 8, width // 4, width // 2]
            cout = [width // 16, width // 8, width // 4, width // 2, width]
            ks = list(_get(custom, "PARALLEL_KERNELS", [3] * 5))
            ps = list(_get(custom, "PARALLEL_PADDINGS", [1] * 5))
            ss = list(_get(custom, "PARALLEL_STRIDES", [2] * 5))
            if ks != [3] * 5 or ps != [1] * 5 or ss[0] != 2:
                raise NotImplementedError("parallel branch kernels/paddings other than 3/1 (or first stride != 2)")
            self.parallel_strides = ss
            self.parallel_branch_v = nn.Sequential(*[
                _ConvBnRelu(cin[j], cout[j], ks[j], ps[j], ss[j]) if j == 0 else
                _ResnetStage(cin[j], cout[j] // 2, cout[j], ks[j], ss[j], ps[j]) for j in range(5)])
            tk = list(_get(custom, "PRALLEL_T2B_KERNELS", None) or [18, 10, 6, 4, 3])
            tp = list(_get(custom, "PRALLEL_T2B_PADDINGS", None) or [1] * 5)
            ts = list(_get(custom, "PRALLEL_T2B_STRIDES", None) or [16, 8, 4, 2, 1])
            if tk != ts or any(tp):
                raise NotImplementedError("lateral adapters need kernel == stride and zero padding (released configs)")
            self.t2b_usecls = bool(_get(custom, "PRALLEL_T2B_USECLS", False))
            self.parallel_lateral_adapter = nn.Sequential(*[
                LateralAdapter(cout[j], width, tk[j], tp[j], ts[j]) for j in range(5)])


class VisualTransformer(nn.Module):
    """M.py:2476-2543.  Two stems: the MS-CLIP-S conv stem inside the Transformer's slot 0 (EARLY_CONV +
    EARLY_CONV_NEW_IMPLEMENT, the released configs) or the plain patch convolution `conv1` (EARLY_CONV off, M.py:2502-2508:
    the reference's only way to a 16 x 16 grid, BASELINE config C5's ViT-L/14)."""

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim, custom):
        super().__init__()
        self.early_conv = bool(_get(custom, "EARLY_CONV", False))
        if self.early_conv and not _get(custom, "EARLY_CONV_NEW_IMPLEMENT", False):
            raise NotImplementedError("EARLY_CONV without EARLY_CONV_NEW_IMPLEMENT (the 5-conv patch stack of M.py:2555-2618) is "
                                      "built by no released config")
        if not self.early_conv:
            if _get(custom, "PARALLEL_IN_V", False):
                raise NotImplementedError("PARALLEL_IN_V without the conv stem: the lateral adapters pool stride-2 maps onto the "
                                          "token grid, which a patch-conv grid does not match (DESIGN.md s9)")
            if input_resolution % patch_size:
                raise NotImplementedError("image size must be a multiple of the patch size")
            self.conv1 = nn.Conv2d(3, width, patch_size, patch_size, bias=False)              # M.py:2502-2508
        self.input_resolution, self.patch_size, self.output_dim = input_resolution, patch_size, output_dim
        self.sequence_length = (input_resolution // patch_size) ** 2 + 1
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.sequence_length, width))
        self.ln_pre = _ln(width)
        self.transformer = Transformer(width, layers, heads, custom, "visual", first_conv=self.early_conv)
        self.ln_post = _ln(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


class CLIP(nn.Module):
    """M.py:2701-2858 / 2979-3155 for the released MS-CLIP-S configs."""

    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 vocab_size, transformer_width, transformer_heads, transformer_layers, gather_tensors=False,
                 custom_config=None, precision="bf16"):
        super().__init__()
        if precision not in ("bf16", "fp8", "fp8-qkv"):
            raise NotImplementedError(f"MODEL.SPEC.PRECISION = {precision!r}: 'bf16' (default), 'fp8' (BASELINE config C5: the MLP "
                                      "projections c_fc / c_proj on the MX fp8 MFMA) or 'fp8-qkv' (in_proj as well: faster, and "
                                      "noisier through the softmax); no reference semantics")
        self.precision = precision
        for key in _UNSUPPORTED_TRUTHY:
            if _get(custom_config, key, False):
                raise NotImplementedError(f"CUSTOM.{key} selects an experimental branch of the reference that no "
                                          f"released MS-CLIP-S config enables; it is not built here")
        if isinstance(custom_config, dict):
            for key, val in custom_config.items():
                if key not in _ACCEPTED and key not in _UNSUPPORTED_TRUTHY and val not in (None, False, 0, 0.0, "", [], ()):
                    raise NotImplementedError(f"CUSTOM.{key}={val!r} is not a switch of the released MS-CLIP-S configs "
                                              f"(accepted keys: {sorted(_ACCEPTED)})")
        if not _get(custom_config, "CUSTOM_ATTN", False):
            raise NotImplementedError("CUSTOM.CUSTOM_ATTN must be True (b32.yaml:59-60)")
        if vision_width % 64 or vision_width != transformer_width:
            raise NotImplementedError("vision and text widths must match (shared blocks) and be multiples of 64")
        self.context_length, self.vocab_size = context_length, vocab_size
        self.transformer_width, self.embed_dim = transformer_width, embed_dim
        self.gather_tensors = gather_tensors
        self.custom_config = custom_config
        vision_heads = vision_width // 64                                   # M.py:2758
        if transformer_width // transformer_heads != 64:
            raise NotImplementedError("head_dim must be 64")
        self.visual = VisualTransformer(image_resolution, vision_patch_size, vision_width, vision_layers, vision_heads,
                                        embed_dim, custom_config)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads, custom_config, "text")
        self.heads = transformer_heads

        shared = _get(custom_config, "SHARE_MODULES", None)
        self.share_from_layer = None
        if shared is not None:                                                # M.py:2786-2830
            n_layers = _get(custom_config, "N_LAYERS", -1)
            self.share_from_layer = max(n_layers, 0) if n_layers != -1 else 0
            for m in shared:
                for i, block in enumerate(self.visual.transformer.resblocks):
                    if n_layers != -1 and i < n_layers:
                        continue
                    if isinstance(block, EarlyconvRes):
                        raise NotImplementedError("sharing slot 0 (the conv stem) is impossible; N_LAYERS must be >= 1")
                    parts = m.split(".")
                    if len(parts) != 1:
                        if parts[0] == "attn":
                            setattr(self.transformer.resblocks[i].attn, parts[1], getattr(block.attn, parts[1]))
                    else:
                        setattr(self.transformer.resblocks[i], m, getattr(block, m))

        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = _ln(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]))                      # T = e, M.py:2850
        self._init_parameters()
        self._engine = None

    # ---- init: trunc_normal(0.02) for Linear/Conv, ones/zeros for norms (M.py:2936-2947, 2853)
    def _init_parameters(self):
        nn.init.trunc_normal_(self.positional_embedding, std=0.02)
        nn.init.trunc_normal_(self.text_projection, std=0.02)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, _Attn):
                nn.init.xavier_uniform_(m.in_proj_weight)                     # M.py:297 (_reset_parameters)
                nn.init.zeros_(m.in_proj_bias)

    # ---- reference surface
    @property
    def dtype(self):
        return self.visual.positional_embedding.dtype                        # M.py:2973-2975

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"positional_embedding", "token_embedding", "logit_scale"}   # M.py:2950-2956

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {}

    # ---- engine lifetime: any parameter movement / reload invalidates the packed weights
    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **k):
        self._engine = None
        return super().load_state_dict(state_dict, strict=strict, **k)

    def train(self, mode=True):
        if mode:
            logging.getLogger(__name__).warning(
                "msclip_amd: forward() / encode_*() are the inference path (no autograd, BatchNorm folded with its running "
                "statistics) whatever the module flags say; the training step -- forward with train-mode BatchNorm, backward "
                "of every parameter, AdamW -- is msclip_amd.train.TrainStep / from_config(model, config)")
        return super().train(mode)

    def engine(self):
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine(self)
        return self._engine

    @torch.no_grad()
    def encode_image(self, image, norm=True, action=None):
        assert action is None
        return self.engine().encode_image(image, norm=norm)

    @torch.no_grad()
    def encode_text(self, text, norm=True, action=None):
        assert action is None
        return self.engine().encode_text(text, norm=norm)

    def stage_captions(self, text):
        """Stage a token batch for encode_text / forward / contrastive_loss (engine.Engine.stage_captions): its per-caption live
        lengths are computed on the device now, so the call that consumes the result never waits for them.  Optional: every
        entry point also takes the plain int64 tensor of the reference API."""
        return self.engine().stage_captions(text)

    @torch.no_grad()
    def forward(self, image, text):
        """logits = exp(logit_scale) * I_all @ T_all^T after the rank-major feature all-gather (M.py:3126-3155).
        Unlike the reference a missing process group means world size 1."""
        return self.engine().forward_logits(image, text, gather=self.gather_tensors)

    @torch.no_grad()
    def contrastive_loss(self, image, text):
        """Symmetric cross-entropy over the global batch (not in the reference)."""
        return self.engine().forward_loss(image, text, gather=self.gather_tensors)


def get_clip_model(config, vocab_size=None, eot_token=None, **kwargs):
    """Same reads as the reference factory (M.py:3182-3227)."""
    spec = config.MODEL.SPEC
    vis, txt = spec.VISION, spec.TEXT
    if vis.MODEL != "vit":
        raise NotImplementedError("ModifiedResNet towers are outside the MS-CLIP-S path")
    if _get(vis, "DROP_PATH", 0.0):
        raise NotImplementedError("DROP_PATH > 0 is a training-time feature")
    if txt.STYLE != "clip" or txt.TOKENIZER != "clip":
        raise NotImplementedError("only the 'clip' text style/tokenizer is built")
    if _get(spec, "POOL_TYPE", "default") != "default" or _get(spec, "SKIP_CLS", False):
        raise NotImplementedError("POOL_TYPE/SKIP_CLS variants are not part of the released configs")
    return CLIP(spec.EMBED_DIM, config.TRAIN.IMAGE_SIZE[0], vis.LAYERS, vis.WIDTH, vis.PATCH_SIZE,
                txt.CONTEXT_LENGTH, vocab_size if vocab_size is not None else txt.VOCAB_SIZE, txt.WIDTH, txt.HEADS,
                txt.LAYERS, gather_tensors=_get(spec, "GATHER_TENSORS", False), custom_config=config.CUSTOM,
                precision=_get(spec, "PRECISION", "bf16"))


build_model = get_clip_model     # the name BASELINE.json's north_star uses; the reference only has get_clip_model

comm = _comm.comm
gather_tensors = _comm.gather_tensors
