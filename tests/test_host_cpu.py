"""Host-side logic that needs no GPU: config surface, checkpoint ABI, weight packing folds (checked against the
oracle), the implicit-GEMM chunk table semantics, the C-ABI library's exports, and the loud failure without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, load_schema, synth_sd
from msclip_amd import hip, packing as P, synth
from msclip_amd.clip_openai_pe_res_v1 import build_model, get_clip_model
from msclip_amd.config import named_config
from oracle import msclip_oracle as O


def test_config_base_inheritance_and_opts():
    c = named_config("b16-yfcc-msclips", ["MODEL.SPEC.EMBED_DIM", "256"])
    assert c.MODEL.SPEC.VISION.PATCH_SIZE == 16 and c.MODEL.SPEC.VISION.WIDTH == 768      # overlay + BASE
    assert c.MODEL.SPEC.TEXT.HEADS == 12 and c.MODEL.SPEC.TEXT.CONTEXT_LENGTH == 77
    assert c.MODEL.SPEC.EMBED_DIM == 256
    assert c.CUSTOM.EARLY_CONV_RES_STRIDES == [2, 2, 2, 1] and c.CUSTOM.CUSTOM_ATTN is True
    with pytest.raises(AttributeError):
        c.NAME = "frozen"


@pytest.mark.parametrize("name", ["b32-yfcc-msclips", "b16-yfcc-msclips", "l16-fp8-msclips", "l14-fp8-msclips"])
def test_state_dict_abi_matches_reference_schema(name):
    model = get_clip_model(named_config(name))
    mine = [(k, tuple(v.shape), v.dtype) for k, v in model.state_dict().items()]
    assert mine == load_schema(name)                       # same 521 keys (l16: 809), order, shapes, dtypes
    # aliases: text block i >= 1 shares attn/mlp tensors with vision block i; LayerNorms never shared
    for i in range(1, len(model.transformer.resblocks)):
        v, t = model.visual.transformer.resblocks[i], model.transformer.resblocks[i]
        assert v.attn.in_proj_weight is t.attn.in_proj_weight and v.attn.in_proj_bias is t.attn.in_proj_bias
        assert v.attn.out_proj is t.attn.out_proj and v.mlp is t.mlp
        assert v.ln_1 is not t.ln_1 and v.ln_2 is not t.ln_2
    assert sum(p.numel() for p in model.parameters()) == {"b32": 132408001, "b16": 132503617, "l16": 369877761, "l14": 368117761}[name[:3]]


def test_strict_load_keeps_aliases_and_build_model_alias():
    name = "b32-yfcc-msclips"
    model = build_model(named_config(name))
    model.load_state_dict(synth_sd(name), strict=True)
    v, t = model.visual.transformer.resblocks[5], model.transformer.resblocks[5]
    assert v.mlp.c_fc.weight.data_ptr() == t.mlp.c_fc.weight.data_ptr()
    assert torch.equal(model.state_dict()["transformer.resblocks.5.attn.in_proj_weight"],
                       synth_sd(name)["visual.transformer.resblocks.5.attn.in_proj_weight"])
    assert model.dtype == torch.float32 and "logit_scale" in model.no_weight_decay()


def test_experimental_switches_are_rejected():
    for key in ("LORA_OPEN", "CONVIT_IN_V", "PARALLEL_B2T", "GUMBEL_SELECT"):
        with pytest.raises(NotImplementedError, match=key):
            get_clip_model(named_config("b32-yfcc-msclips", [f"CUSTOM.{key}", "True"]))
    with pytest.raises(NotImplementedError):
        get_clip_model(named_config("b32-yfcc-msclips", ["CUSTOM.EARLY_CONV", "False"]))


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    model = get_clip_model(named_config("b32-yfcc-msclips"))
    with pytest.raises(hip.HipUnavailable):
        model.encode_text(torch.zeros(1, 77, dtype=torch.long))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "msclip_hip.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    declared = set(re.findall(r"\b(msclip_[a-z0-9_]+)\s*\(", text))
    assert declared == set(hip.EXPORTS)
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.msclip_abi_version.restype = ctypes.c_int
    lib.msclip_build_arch.restype = ctypes.c_char_p
    assert lib.msclip_abi_version() == hip.ABI_VERSION == 8 and lib.msclip_build_arch() == b"gfx950"
    # struct mirror must match the C layout (6 pointers + 24 ints/floats, then a pointer in the middle)
    assert ctypes.sizeof(hip.GemmDesc) % 8 == 0 and hip.GemmDesc.ktab.offset % 8 == 0


def test_entry_points_reject_bad_arguments_before_launching():
    """Argument validation of the C ABI runs on the host, before any launch: null pointers, unsupported channel
    counts, sizes whose byte offsets would not fit the kernels' 32-bit buffer addressing (include/msclip_hip.h)."""
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    EINVAL = -1
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, vp)
    lib.msclip_conv1x1_conv3x3s2.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    assert lib.msclip_conv1x1_conv3x3s2(None, p, p, p, p, p, 1, 8, 8, 48, None) == EINVAL          # null input
    assert lib.msclip_conv1x1_conv3x3s2(p, p, p, p, p, p, 1, 8, 8, 64, None) == EINVAL             # Cout not 48 / 96
    assert lib.msclip_conv1x1_conv3x3s2(p, p, p, p, p, p, 4096, 112, 112, 48, None) == EINVAL      # input >= 2 GiB
    lib.msclip_convresblock48_s2.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]
    assert lib.msclip_convresblock48_s2(p, p, p, p, p, p, None, p, p, 1, 8, 8, None) == EINVAL     # null shortcut weights
    assert lib.msclip_convresblock48_s2(p, p, p, p, p, p, p, p, p, 0, 8, 8, None) == EINVAL        # empty batch
    lib.msclip_stem_dual_conv3x3s2.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    assert lib.msclip_stem_dual_conv3x3s2(p, 0, p, p, p, p, p, p, 1, 224, 224, 32, None) == EINVAL
    assert lib.msclip_stem_dual_conv3x3s2(p, 0, p, p, p, p, p, p, 4000, 224, 224, 96, None) == EINVAL   # image >= 2 GiB
    lib.msclip_layernorm_split.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, ctypes.c_float, vp]
    assert lib.msclip_layernorm_split(p, 768, p, p, p, p, 9, p, 768, 0, 8, 768, 1e-12, None) == EINVAL   # split > M
    assert lib.msclip_layernorm_split(p, 770, p, p, p, p, 4, p, 768, 0, 8, 768, 1e-12, None) == EINVAL   # ldx % 4
    lib.msclip_adapter_combine_ln.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ctypes.c_float, vp]
    assert lib.msclip_adapter_combine_ln(p, 768, p, 768, p, p, p, p, p, 768, 1, 51, 7, 768, 1, 1e-12, None) == EINVAL   # L != g*g+1
    assert lib.msclip_adapter_combine_ln(p, 768, p, 768, p, p, p, p, p, 768, 1, 50, 7, 768, 1, 1e-12, None) == EINVAL   # in place
    lib.msclip_dwpool.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    assert lib.msclip_dwpool(p, p, p, 48, 1, 112, 112, 48, 5, None) == EINVAL                      # H % k
    lib.msclip_gather_rows.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    assert lib.msclip_gather_rows(None, 3072, None, 1, 0, p, 3072, 4, 3072, None) == EINVAL        # null input
    assert lib.msclip_gather_rows(p, 3072, None, 1, 0, p, 3072, 4, 3000, None) == EINVAL           # rows not 16-byte pieces
    assert lib.msclip_gather_rows(p, 3072, None, 1, 0, p, 3072, 0, 3072, None) == EINVAL           # no rows
    lib.msclip_gemm.argtypes = [vp, vp]
    assert lib.msclip_gemm(None, None) == EINVAL


# ---------------------------------------------------------------- packing folds vs oracle (CPU, fp32 weights)

def _run_spec(x_nchw, spec, relu=False, resid=None):
    """Emulate the gathering GEMM with fp32 arithmetic on the bf16-rounded packed weight."""
    w = spec.weight.float()[:, :spec.kh * spec.kw * spec.cin].reshape(spec.cout, spec.kh, spec.kw, spec.cin)
    y = F.conv2d(x_nchw, w.permute(0, 3, 1, 2), stride=spec.stride, padding=spec.pad) + spec.bias[None, :, None, None]
    if resid is not None:
        y = y + resid
    return F.relu(y) if relu else y


def test_stem_and_parallel_folds_match_oracle():
    name, arch = "b32-yfcc-msclips", O.arch_b32()
    sd = synth_sd(name)
    img = synth.synth_images(1, seed=3)[:, :, :64, :64]
    # dual first conv
    w, b = P.stem_dual_weights(sd, "visual.transformer.resblocks.0", "visual.transformer.parallel_branch_v.0")
    y = F.relu(F.conv2d(img, w.t().reshape(96, 3, 3, 3), stride=2, padding=1) + b[None, :, None, None])
    taps = {}
    sp = "visual.transformer.resblocks.0"
    s1 = F.relu(O.batch_norm(F.conv2d(img, sd[sp + ".conv1.weight"], stride=2, padding=1), sd, sp + ".bn1", 1e-5))
    p0 = O.parallel_stage(img, sd, arch, 0)
    assert (y[:, :48] - s1).abs().max() < 1e-5 and (y[:, 48:] - p0).abs().max() < 1e-5
    # stem stage 0 with the shortcut merged into the centre tap (weights are bf16-rounded: loose tolerance)
    spec = P.stem_stage(sd, sp + ".resnet_stage.conv_0", 32, 2)
    q = sp + ".resnet_stage.conv_0"
    ref = F.relu(O.batch_norm(F.conv2d(s1, sd[q + ".conv1.weight"], stride=2, padding=1), sd, q + ".bn1", 1e-5) +
                 O.batch_norm(F.conv2d(s1, sd[q + ".downsample.0.weight"], stride=2), sd, q + ".downsample.1", 1e-5))
    got = _run_spec(s1, spec, relu=True)
    assert got.shape == ref.shape and (got - ref).abs().max() < 3e-2 * ref.abs().max()
    assert (got - ref).abs().mean() < 3e-3 * ref.abs().mean() + 1e-4
    # bottleneck stage 1
    c1, c2, cr, c3 = P.bottleneck(sd, "visual.transformer.parallel_branch_v.1.resnet_stage.conv_0", 32, 2)
    y1 = _run_spec(p0, c1, relu=True)
    y2 = _run_spec(y1, c2, relu=True)
    out = _run_spec(y2, c3, relu=True, resid=_run_spec(p0, cr))
    ref = O.parallel_stage(p0, sd, arch, 1)
    assert out.shape == ref.shape and (out - ref).abs().mean() < 5e-3 * ref.abs().mean() + 1e-4


def test_adapter_fold_matches_oracle():
    name, arch = "b32-yfcc-msclips", O.arch_b32()
    sd = synth_sd(name)
    g = arch.grid
    torch.manual_seed(0)
    j = 2
    top = torch.randn(2, 192, g * 4, g * 4)
    x = torch.randn(2, g * g + 1, 768)
    ref = O.lateral_adapter(top, x, sd, arch, j)
    pool, k, pw, dww, dwb = P.adapter_weights(sd, f"visual.transformer.parallel_lateral_adapter.{j}", g)
    assert k == 4
    pooled = F.conv2d(top, pool.t().reshape(192, 1, k, k), stride=k, groups=192)
    t = _run_spec(pooled, pw).flatten(2).transpose(1, 2)
    grid = x[:, 1:].transpose(1, 2).reshape(2, 768, g, g)
    bo = (F.conv2d(grid, dww.t().reshape(768, 1, 3, 3), padding=1, groups=768) + dwb[None, :, None, None])
    v = torch.cat([2 * x[:, :1], bo.flatten(2).transpose(1, 2) + t], 1)
    p = f"visual.transformer.parallel_lateral_adapter.{j}"
    got = O.layer_norm(v, sd[p + ".ln_adapt.weight"], sd[p + ".ln_adapt.bias"])
    assert (got - ref).abs().max() < 2e-2


def test_ktab_semantics_tiny_conv():
    """Interpret the chunk table exactly like gemm.hip's loader and compare with F.conv2d."""
    torch.manual_seed(1)
    B, H, Cin, Cout, stride, pad = 2, 6, 16, 8, 2, 1
    x = torch.randn(B, Cin, H, H)
    w = torch.randn(Cout, Cin, 3, 3)
    spec = P.ConvSpec(w, torch.zeros(Cout), H, H, stride, pad)
    xn = x.permute(0, 2, 3, 1).contiguous().flatten().numpy()           # NHWC
    tab = spec.ktab.numpy()
    Ho = spec.h_out
    rows = np.zeros((B * Ho * Ho, spec.weight.shape[1]), dtype=np.float32)
    for m in range(B * Ho * Ho):
        b, p = divmod(m, Ho * Ho)
        ho, wo = divmod(p, Ho)
        ih0, iw0 = ho * stride - pad, wo * stride - pad
        pix = ((b * H + ih0) * H + iw0) * Cin
        for c, e in enumerate(tab):
            if e < 0:
                continue
            kh, kw, doff = (e >> 20) & 15, (e >> 24) & 15, e & 0xFFFFF
            if 0 <= ih0 + kh < H and 0 <= iw0 + kw < H:
                rows[m, c * 8:c * 8 + 8] = xn[pix + doff:pix + doff + 8]
    got = torch.from_numpy(rows) @ spec.weight.float().t()
    ref = F.conv2d(x, w.to(torch.bfloat16).float(), stride=stride, padding=pad).permute(0, 2, 3, 1).reshape(-1, Cout)
    assert (got - ref).abs().max() < 1e-3
    assert spec.weight.shape[1] % 64 == 0 and (tab[(9 * Cin) // 8:] < 0).all()


def test_qkv_prescale_is_exact():
    torch.manual_seed(0)
    w, b = torch.randn(2304, 768), torch.randn(2304)
    wq, bq = P.qkv_weights(w, b, 12)
    assert torch.equal(wq[:768].float(), (w[:768].to(torch.bfloat16).float() * 0.125))
    assert torch.equal(wq[768:], w[768:].to(torch.bfloat16)) and torch.equal(bq[:768], b[:768] * 0.125)


def test_gemm_dispatch_rule_is_the_librarys_own():
    """msclip_gemm_variant (the dispatch rule msclip_gemm itself follows, csrc/gemm.hip::pick_variant) for the shapes of
    the B/32 step at batch 512: bench.py counts the dominant kernel's launches by this name.  No GPU work."""
    from msclip_amd import hip

    def v(mode, M, N, tile, K, conv=None, **kw):
        return hip.gemm_variant(hip.describe_gemm(mode, M, N, K, tile, conv, **kw))
    c3 = lambda B, H, C: (H, H, C, H // 2, H // 2, 2, 1)
    assert v(0, 65024, 2304, 0, 768) == "pp"            # QKV over image + text rows
    assert v(0, 65024, 768, 0, 3072) == "pp"            # c_proj
    assert v(0, 65024, 768, 4, 768) == "pp"
    assert v(0, 65024, 768, 2, 768) == "invalid"        # retired main loop
    assert v(0, 65024, 2304, 7, 768) == "invalid"       # rounds 2-3's epilogue-hiding kernels: retired in round 4
    assert v(0, 65024, 2304, 8, 768) == "invalid"
    assert v(0, 65024, 2304, 9, 768) == "invalid"
    assert v(0, 6422528, 48, 0, 64, ldx=48) == "stream"  # pointwise conv of the conv branch
    assert v(0, 25088, 768, 0, 192) == "stream"         # adapter 1x1
    assert v(0, 512, 512, 0, 768) == "dense128"         # heads
    assert v(1, 1605632, 96, 0, 448, c3(512, 112, 48)) == "stream"        # 3x3 stride 2, 48 input channels
    assert v(1, 401408, 192, 0, 896, c3(512, 56, 96)) == "conv192"        # 3x3 stride 2, 96 -> 192
    assert v(1, 100352, 384, 0, 1728, c3(512, 28, 192)) == "ppconv"       # 3x3 stride 2, 192 -> 384 (K-tile inside one tap)
    assert v(1, 401408, 96, 0, 896, c3(512, 56, 96)) == "conv128"
    assert v(1, 1000, 64, 0, 576, (10, 10, 64, 10, 10, 1, 1)) == "conv128"
    assert v(0, 65024, 768, 0, 100) == "invalid"        # K % 64
    assert v(0, 25088, 768, 0, 64, rpg=49) == "stream"  # plain row scatter streams too (round 4: the parity-class input gradients)
    assert v(0, 25088, 768, 0, 768, rpg=49) == "pp"


def test_reference_shim_layout_import(tmp_path):
    """the reference imports the model as `models.clip_openai_pe_res_v1` with its lib/ directory on
    sys.path (tools/_init_paths.py:14-17, tools/zero_shot.py:40).  A lib/models/clip_openai_pe_res_v1.py holding only
    the documented re-export must give the reference's call sequence (factory, strict load, attribute names)."""
    import subprocess
    import sys
    lib = tmp_path / "lib" / "models"
    lib.mkdir(parents=True)
    (lib / "__init__.py").write_text("")
    (lib / "clip_openai_pe_res_v1.py").write_text(
        "from msclip_amd.clip_openai_pe_res_v1 import (   # noqa: F401\n"
        "    CLIP, get_clip_model, build_model, comm, gather_tensors)\n")
    code = f"""
import sys
sys.path.insert(0, {str(tmp_path / 'lib')!r})          # what tools/_init_paths.py does
sys.path.insert(0, {ROOT!r})
from models import clip_openai_pe_res_v1
from msclip_amd.config import named_config
config = named_config("b32-yfcc-msclips")
model = clip_openai_pe_res_v1.get_clip_model(config)
assert type(model).__name__ == "CLIP" and len(model.state_dict()) == 521
for attr in ("logit_scale", "visual", "transformer", "token_embedding", "positional_embedding", "text_projection", "ln_final"):
    assert hasattr(model, attr), attr
assert clip_openai_pe_res_v1.comm.world_size == 1 and callable(clip_openai_pe_res_v1.gather_tensors)
print("shim ok")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "shim ok" in r.stdout, r.stderr[-2000:]


def test_packaged_text_data_and_eval_cli_config():
    """The BPE merges and the ImageNet prompts ship with the package (f1 / f2 run stand-alone on the GPU box), and the
    eval CLI builds its config like the reference (tools/zero_shot.py:183-190)."""
    import sys
    from msclip_amd import zeroshot
    from msclip_amd.tokenizer import SimpleTokenizer, find_vocab
    assert os.path.isfile(find_vocab()) and "msclip_amd" in find_vocab()
    tok = SimpleTokenizer()
    assert tok.get_vocab_size() == 49408
    classes, templates = zeroshot.load_prompts("imagenet")
    assert len(classes) == 1000 and len(templates) == 80 and classes[0] == "tench" and all("{}" in t for t in templates)
    with pytest.raises(ValueError, match="Can not find prompt"):
        zeroshot.load_prompts("no-such-dataset")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import eval_zeroshot as E
    a = E.parse_args(["--ds", "imagenet", "--model", os.path.join(ROOT, "experiments/model/b16-yfcc-msclips.yaml"),
                      "DATASET.ROOT", "/data/in1k/", "TEST.BATCH_SIZE_PER_GPU", "64"])
    c = E.build_config(E.resolve_dataset("imagenet"), a.model, a.opts)
    assert c.NAME == "" and c.DATASET.ROOT == "/data/in1k/" and c.DATASET.TEST_SET == "val" and c.TEST.METRIC == "accuracy"
    assert c.TEST.BATCH_SIZE_PER_GPU == 64 and c.MODEL.SPEC.VISION.PATCH_SIZE == 16
    assert c.MODEL.PRETRAINED_MODEL.endswith("b16-yfcc-msclips_ckpt.pth")
    with pytest.raises(Exception, match="does not exist"):
        E.resolve_dataset("cifar-1000")


def test_update_config_follows_reference_naming_and_lr_scaling():
    """lib/config/default.py:294-306: NAME = file name + NAME (prefix); TRAIN.LR and CUSTOM.LR_SHARE scale with the
    world size when TRAIN.SCALE_LR (world size 1 here: unchanged)."""
    c = named_config("b32-yfcc-msclips")
    assert c.NAME == "b32-yfcc-msclips"
    assert abs(c.TRAIN.LR - 1e-4) < 1e-12 and abs(c.CUSTOM.LR_SHARE - 1e-4) < 1e-12 and c.CUSTOM.WD_SHARE == 0.2


def test_optimizer_block_matches_reference_yaml_literals():
    """TRAIN / CUSTOM optimizer keys as the reference yaml spells them (experiments/model/b32.yaml:32-52,
    b32-yfcc-msclips.yaml:13-14).  Expected values are LITERALS copied from those files, not read back from the config."""
    from msclip_amd import train
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    cfg = named_config("b32-yfcc-msclips")
    assert cfg.TRAIN.OPTIMIZER == "adamW" and cfg.TRAIN.WD == 0.05 and cfg.TRAIN.LR == 0.0001
    assert list(cfg.TRAIN.WITHOUT_WD_LIST) == ["bn", "bias", "ln"]
    assert cfg.TRAIN.LR_SCHEDULER.METHOD == "timm" and cfg.TRAIN.LR_SCHEDULER.ARGS.sched == "cosine"
    assert cfg.TRAIN.LR_SCHEDULER.ARGS.warmup_epochs == 5 and cfg.TRAIN.LR_SCHEDULER.ARGS.min_lr == 0.00001
    st = train.optimizer_settings(cfg)
    assert st == dict(lr=0.0001, lr_share=0.0001, wd=0.05, wd_share=0.2, betas=(0.9, 0.999), eps=1e-8,
                      without_wd=("bn", "bias", "ln"))
    m = get_clip_model(cfg)
    g = {k: (lr, wd) for k, _, lr, wd in train.param_groups(m, st["lr"], st["lr_share"], st["wd"], st["wd_share"], st["without_wd"])}
    assert g["visual.transformer.resblocks.3.mlp.c_fc.weight"] == (0.0001, 0.2)          # shared: LR_SHARE / WD_SHARE
    assert g["visual.transformer.resblocks.3.attn.in_proj_weight"] == (0.0001, 0.2)
    assert g["transformer.resblocks.0.mlp.c_fc.weight"] == (0.0001, 0.05)                # text block 0 is not shared: TRAIN.WD
    assert g["visual.transformer.resblocks.0.conv1.weight"] == (0.0001, 0.05)
    assert g["visual.proj"] == (0.0001, 0.05) and g["text_projection"] == (0.0001, 0.05)
    for k in ("visual.transformer.resblocks.3.mlp.c_fc.bias", "visual.transformer.resblocks.3.attn.in_proj_bias",     # 'bias'
              "visual.transformer.resblocks.3.ln_1.weight", "ln_final.weight", "visual.ln_pre.weight",                # 'ln'
              "visual.transformer.parallel_lateral_adapter.2.ln_adapt.weight",
              "visual.transformer.resblocks.0.bn1.weight",                                                            # 'bn'
              "visual.transformer.resblocks.0.resnet_stage.conv_1.downsample.1.weight",        # a BatchNorm without 'bn' in its name
              "visual.transformer.parallel_branch_v.2.resnet_stage.conv_0.residual_bn.weight",
              "visual.positional_embedding", "positional_embedding", "token_embedding.weight", "logit_scale"):        # no_weight_decay()
        assert g[k][1] == 0.0, k
    # an optimizer this build does not implement is an error, not a silent AdamW
    sgd = named_config("b32-yfcc-msclips", ["TRAIN.OPTIMIZER", "sgd"])
    with pytest.raises(NotImplementedError):
        train.optimizer_settings(sgd)
    # OPTIMIZER_ARGS override betas / eps
    oa = named_config("b32-yfcc-msclips", ["TRAIN.OPTIMIZER_ARGS.betas", "[0.9, 0.98]", "TRAIN.OPTIMIZER_ARGS.eps", "1e-6"])
    st2 = train.optimizer_settings(oa)
    assert st2["betas"] == (0.9, 0.98) and st2["eps"] == 1e-6


def test_lr_schedule_matches_timm_cosine_with_the_reference_yaml_literals():
    """TRAIN.LR_SCHEDULER of the reference yaml (experiments/model/b32.yaml:40-48: timm cosine, warm-up 5 epochs from 1e-6,
    floor 1e-5, 10 cool-down epochs; lib/config/default.py:306-308 sets ARGS.epochs = END_EPOCH = 50) through
    train.lr_schedule: expected values are timm's CosineLRScheduler formula evaluated by hand on the yaml's literals."""
    import math
    from msclip_amd import train
    sch = train.lr_schedule(named_config("b32-yfcc-msclips"))
    assert (sch.epochs, sch.warmup_epochs, sch.warmup_lr, sch.min_lr, sch.cooldown_epochs, sch.total_epochs) == (50, 5, 1e-6, 1e-5, 10, 60)
    base = 1e-4
    assert sch.lr_at(base, 0) == 1e-6                                            # warm-up starts at warmup_lr ...
    assert abs(sch.lr_at(base, 3) - (1e-6 + 3 * (1e-4 - 1e-6) / 5)) < 1e-12      # ... and climbs linearly
    assert abs(sch.lr_at(base, 5) - (1e-5 + 0.5 * 9e-5 * (1 + math.cos(math.pi * 5 / 50)))) < 1e-12   # no warm-up prefix: t is the epoch
    assert abs(sch.lr_at(base, 25) - (1e-5 + 0.5 * 9e-5)) < 1e-12                # half way down the cosine
    assert sch.lr_at(base, 50) == 1e-5 and sch.lr_at(base, 59) == 1e-5           # cool-down at the floor
    assert abs(sch.lr_at(0.0016, 25) - (1e-5 + 0.5 * (0.0016 - 1e-5))) < 1e-12   # each group's own base rate (CUSTOM.LR_SHARE)
    cfg = named_config("b32-yfcc-msclips", ["TRAIN.LR_SCHEDULER.METHOD", "MultiStep"])
    with pytest.raises(NotImplementedError):
        train.lr_schedule(cfg)


def test_plan_executor_records_and_replays_without_a_gpu():
    """The native step executor (include/msclip_hip.h "Launch plans") on the host side only: an entry point called while a plan
    records on this thread is appended to the table BEFORE it validates its arguments, so a call that is rejected (null
    pointers: no launch, no GPU needed) is still recorded and its replay returns the same rejection; a call on a stream that is
    not one of the plan's slots makes msclip_plan_end fail; nothing is recorded outside begin / end."""
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    L = hip.lib()
    vp = ctypes.c_void_p
    s1, s2, s3 = vp(0x1000), vp(0x2000), vp(0x3000)

    def finalize(stream):
        return L.msclip_rowstat_finalize(None, 12, None, None, 256, 768, 1e-12, None, stream)

    h = vp()
    assert L.msclip_plan_create(ctypes.byref(h)) == 0
    assert finalize(s1) == -1 and L.msclip_plan_size(h) == 0            # not recording: nothing appended
    assert L.msclip_plan_begin(h, (vp * 2)(s1, s2), 2) == 0
    assert finalize(s1) == -1 and finalize(s2) == -1                    # rejected, but recorded
    assert L.msclip_plan_size(h) == 2
    assert L.msclip_plan_end(h) == 2
    assert [L.msclip_plan_op_name(h, i) for i in range(2)] == [b"msclip_rowstat_finalize"] * 2
    n = [ctypes.c_int() for _ in range(5)]
    assert L.msclip_plan_info(h, *[ctypes.byref(x) for x in n]) == 0
    assert [x.value for x in n] == [2, 2, 0, 2, 0]
    assert finalize(s1) == -1 and L.msclip_plan_size(h) == 2            # recording is over
    assert L.msclip_plan_run(h, (vp * 2)(s3, s1), 2, None, 0) == -1     # the replayed entry point's own verdict
    assert L.msclip_plan_run(h, (vp * 1)(s3), 1, None, 0) == -1         # wrong number of stream slots
    assert L.msclip_plan_destroy(h) == 0
    h = vp()
    assert L.msclip_plan_create(ctypes.byref(h)) == 0
    assert L.msclip_plan_begin(h, (vp * 1)(s1), 1) == 0
    finalize(s3)                                                        # a stream the plan does not know
    assert L.msclip_plan_end(h) == -1
    assert L.msclip_plan_destroy(h) == 0
