"""ctypes binding of libmsclip_hip.so (the C ABI declared in include/msclip_hip.h).

The library is built in-tree by msclip_amd/csrc/build.sh (hipcc, gfx950) and is
the ONLY compute path of the product: there is no CPU or eager-PyTorch
fallback.  Anything that needs a kernel raises HipUnavailable when the
library is missing or no GPU is visible.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSCLIP_HIP_LIB") or os.path.join(_HERE, "csrc", "libmsclip_hip.so")   # override: kernel A/B probes only
INT_MAX = 2 ** 31 - 1
ABI_VERSION = 8                                          # include/msclip_hip.h MSCLIP_ABI_VERSION

EXPORTS = (
    "msclip_gemm", "msclip_gemm_f8", "msclip_layernorm_stats", "msclip_rowstat_finalize", "msclip_layernorm_f8", "msclip_quant_f8_rows", "msclip_gemm_variant", "msclip_attention", "msclip_attention_lastq", "msclip_layernorm", "msclip_layernorm_split", "msclip_embed_tokens", "msclip_fill_cls",
    "msclip_adapter_combine_ln", "msclip_adapter_combine_ln_stats", "msclip_l2norm", "msclip_gather_rows", "msclip_stem_conv3x3s2_dual", "msclip_stem_conv3x3s2_dual_raw", "msclip_stem_conv3x3s2_dual_stats", "msclip_stem_conv3x3s2_dual_norm", "msclip_dwpool",
    "msclip_stem_dual_conv3x3s2", "msclip_conv1x1_conv3x3s2", "msclip_convresblock48_s2", "msclip_patchify",
    "msclip_lse_rows", "msclip_clip_loss_partial", "msclip_clip_lse_fused", "msclip_clip_loss_from_partials",
    "msclip_transpose_bf16", "msclip_cast_bf16", "msclip_cast_bf16_colsum", "msclip_colsum", "msclip_quickgelu", "msclip_quickgelu_bwd", "msclip_layernorm_bwd",
    "msclip_attention_bwd", "msclip_l2norm_bwd", "msclip_clip_loss_bwd_g", "msclip_embed_tokens_bwd", "msclip_adapter_sum",
    "msclip_adapter_dx", "msclip_adamw", "msclip_adamw_multi", "msclip_im2col", "msclip_col2im", "msclip_relu_bwd", "msclip_dwpool_bwd",
    "msclip_dwpool_wgrad", "msclip_dw3x3_wgrad", "msclip_gemm_splitk", "msclip_gemm_splitk_tn", "msclip_bn_stats", "msclip_bn_apply",
    "msclip_bn_bwd_reduce", "msclip_bn_bwd_dx", "msclip_bn_bwd_fused", "msclip_bn_fold_bwd", "msclip_bn_finish", "msclip_bn_finish_tiled", "msclip_bn_bwd_finish",
    "msclip_text_lengths", "msclip_embed_tokens_packed", "msclip_attention_varlen", "msclip_attention_lastq_varlen",
    "msclip_attention_bwd_varlen", "msclip_embed_tokens_bwd_packed",
    "msclip_qkv_attention", "msclip_qkvattn_tables", "msclip_pack_weights", "msclip_transpose_bf16_multi", "msclip_image_conv_wgrad", "msclip_colsum_multi",
    "msclip_abi_version", "msclip_build_arch",
    "msclip_stream_priority_range", "msclip_stream_create", "msclip_stream_destroy", "msclip_stream_create_cu_masked",
    "msclip_plan_create", "msclip_plan_destroy", "msclip_plan_begin", "msclip_plan_bind_external", "msclip_plan_record_event",
    "msclip_plan_wait_event", "msclip_plan_end", "msclip_plan_abort", "msclip_plan_info", "msclip_plan_op_name", "msclip_plan_run",
    "msclip_plan_size", "msclip_plan_probe_enable", "msclip_plan_probe_disable", "msclip_plan_probe_runs", "msclip_plan_probe_elapsed",
    "msclip_comm_unique_id", "msclip_comm_init", "msclip_comm_destroy", "msclip_comm_async_error", "msclip_allgather_feats",
    "msclip_allreduce", "msclip_prepare_device",
)


class HipUnavailable(RuntimeError):
    pass


class HipError(RuntimeError):
    pass


class BnBwdSide(ctypes.Structure):
    """Mirror of struct msclip_bn_bwd_side."""
    _fields_ = [("x", ctypes.c_void_p), ("ld", ctypes.c_int), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p),
                ("gamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dx", ctypes.c_void_p),
                ("lddx", ctypes.c_int), ("part", ctypes.c_void_p), ("x_bf16", ctypes.c_int)]


class GemmDesc(ctypes.Structure):
    """Mirror of struct msclip_gemm_desc."""
    _fields_ = [
        ("X", ctypes.c_void_p), ("W", ctypes.c_void_p), ("zero", ctypes.c_void_p),
        ("out", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("resid", ctypes.c_void_p),
        ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
        ("ldx", ctypes.c_int), ("ldw", ctypes.c_int), ("ldo", ctypes.c_int), ("ldr", ctypes.c_int),
        ("mode", ctypes.c_int),
        ("H", ctypes.c_int), ("Wd", ctypes.c_int), ("Cin", ctypes.c_int), ("Ho", ctypes.c_int),
        ("Wo", ctypes.c_int), ("stride", ctypes.c_int), ("pad", ctypes.c_int),
        ("ktab", ctypes.c_void_p),
        ("act", ctypes.c_int), ("resid_kind", ctypes.c_int), ("out_kind", ctypes.c_int),
        ("alpha", ctypes.c_float),
        ("rpg", ctypes.c_int), ("radd", ctypes.c_int), ("roff", ctypes.c_int), ("out2", ctypes.c_void_p), ("out_scale", ctypes.c_float), ("tile", ctypes.c_int), ("wg_cap", ctypes.c_int),
        # LayerNorm fold: consumer (W2, bias2, csum, csum2, rowstat, seg_split) and producer (ldxb, xb, center, part)
        ("W2", ctypes.c_void_p), ("bias2", ctypes.c_void_p), ("csum", ctypes.c_void_p), ("csum2", ctypes.c_void_p),
        ("rowstat", ctypes.c_void_p), ("seg_split", ctypes.c_int), ("ldxb", ctypes.c_int), ("xb", ctypes.c_void_p),
        ("center", ctypes.c_void_p), ("part", ctypes.c_void_p), ("resid2", ctypes.c_void_p),
        ("M_dev", ctypes.c_void_p),                  # device int: the kernel runs min(M, *M_dev) rows (packed captions)
        ("bn_mode", ctypes.c_int), ("part_rows", ctypes.c_int), ("bn_consts", ctypes.c_void_p),   # two-pass train-mode BatchNorm
    ]


class QkvAttnDesc(ctypes.Structure):
    """Mirror of struct msclip_qkvattn_desc."""
    _fields_ = [
        ("X", ctypes.c_void_p), ("W", ctypes.c_void_p), ("zero", ctypes.c_void_p), ("out", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("cu", ctypes.c_void_p), ("tile_first", ctypes.c_void_p), ("rowseg", ctypes.c_void_p),
        ("ntiles_dev", ctypes.c_void_p),
        ("M", ctypes.c_int), ("K", ctypes.c_int), ("heads", ctypes.c_int), ("ntiles", ctypes.c_int),
        ("ldx", ctypes.c_int), ("ldw", ctypes.c_int), ("ldo", ctypes.c_int), ("causal_from_row", ctypes.c_int),
        ("rowstat", ctypes.c_void_p), ("csum", ctypes.c_void_p), ("W2", ctypes.c_void_p), ("bias2", ctypes.c_void_p),
        ("csum2", ctypes.c_void_p), ("seg_split", ctypes.c_int),
    ]


_lib = None


def build(force=False):
    """Compile the library in-tree (used by __graft_entry__.build)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["bash", script], check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipUnavailable(f"{LIB_PATH} is missing: run msclip_amd/csrc/build.sh (no CPU fallback exists)")
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.msclip_gemm.argtypes = [ctypes.POINTER(GemmDesc), vp]
        L.msclip_gemm_splitk.argtypes = [ctypes.POINTER(GemmDesc), ci, vp]
        L.msclip_gemm_splitk_tn.argtypes = [ctypes.POINTER(GemmDesc), ci, vp]
        L.msclip_gemm_f8.argtypes = [ctypes.POINTER(GemmDesc), vp, vp, vp]
        L.msclip_layernorm_f8.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, ci, cf, vp, vp]
        L.msclip_quant_f8_rows.argtypes = [vp, ci, vp, ci, vp, ci, ci, vp]
        L.msclip_gemm_variant.argtypes = [ctypes.POINTER(GemmDesc)]
        L.msclip_gemm_variant.restype = ctypes.c_char_p
        L.msclip_attention.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.msclip_attention_lastq.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, ci, vp]
        L.msclip_layernorm.argtypes = [vp, ci, vp, ci, ci, vp, vp, vp, ci, ci, vp, ci, ci, ci, cf, vp]
        L.msclip_layernorm_stats.argtypes = [vp, ci, vp, vp, vp, ci, ci, vp, ci, vp, vp, ci, ci, cf, vp, vp]
        L.msclip_rowstat_finalize.argtypes = [vp, ci, vp, vp, ci, ci, cf, vp, vp]
        L.msclip_layernorm_split.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, cf, vp]
        L.msclip_embed_tokens.argtypes = [vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, vp]
        L.msclip_fill_cls.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
        L.msclip_adapter_combine_ln.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, vp]
        L.msclip_adapter_combine_ln_stats.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, ci, ci, ci, ci, ci, cf, vp]
        L.msclip_l2norm.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci, vp]
        L.msclip_gather_rows.argtypes = [vp, ctypes.c_longlong, vp, ci, ci, vp, ctypes.c_longlong, ci, ci, vp]
        L.msclip_stem_conv3x3s2_dual.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, vp]
        L.msclip_stem_conv3x3s2_dual_raw.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, vp]
        L.msclip_stem_conv3x3s2_dual_stats.argtypes = [vp, ci, vp, vp, ci, ci, ci, ci, vp]
        L.msclip_stem_conv3x3s2_dual_norm.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]
        L.msclip_dwpool.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.msclip_stem_dual_conv3x3s2.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
        L.msclip_conv1x1_conv3x3s2.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
        L.msclip_convresblock48_s2.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]
        L.msclip_patchify.argtypes = [vp, ci, vp, ci, ci, ci, ci, ci, vp]
        L.msclip_lse_rows.argtypes = [vp, ci, vp, ci, ci, vp]
        L.msclip_clip_loss_partial.argtypes = [vp, vp, vp, ci, ci, ci, cf, vp, vp]
        L.msclip_clip_lse_fused.argtypes = [vp, ci, vp, ci, ci, ci, ci, cf, ci, ci, vp, vp, vp, vp]
        L.msclip_clip_loss_from_partials.argtypes = [vp, vp, vp, vp, vp, ci, ci, cf, vp, vp, vp]
        ll = ctypes.c_longlong
        L.msclip_transpose_bf16.argtypes = [vp, ci, vp, ci, ci, ci, ci, vp]
        L.msclip_cast_bf16.argtypes = [vp, ci, vp, ci, ci, ci, vp]
        L.msclip_cast_bf16_colsum.argtypes = [vp, ci, vp, ci, ci, ci, vp, ci, ci, vp]
        L.msclip_colsum.argtypes = [vp, ci, ci, vp, ci, ci, ci, vp, ci, vp]
        L.msclip_quickgelu.argtypes = [vp, vp, ll, vp]
        L.msclip_quickgelu_bwd.argtypes = [vp, vp, vp, ll, vp]
        L.msclip_layernorm_bwd.argtypes = [vp, ci, vp, ci, vp, ci, ci, vp, vp, ci, ci, vp, ci, ci, ci, cf, vp, ci, vp, ci, vp]
        L.msclip_attention_bwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp]
        L.msclip_l2norm_bwd.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci, vp]
        L.msclip_clip_loss_bwd_g.argtypes = [vp, ci, vp, vp, ci, cf, vp, ci, vp, ci, ci, ci, vp]
        L.msclip_embed_tokens_bwd.argtypes = [vp, vp, ci, vp, vp, ci, ci, ci, ci, vp]
        L.msclip_adapter_sum.argtypes = [vp, ci, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.msclip_adapter_dx.argtypes = [vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.msclip_adamw.argtypes = [vp, vp, vp, vp, ll, cf, cf, cf, cf, cf, ci, vp]
        L.msclip_adamw_multi.argtypes = [ctypes.POINTER(AdamwTensor), ci, cf, cf, cf, ci, vp]
        L.msclip_im2col.argtypes = [vp, ci, vp] + [ci] * 11 + [vp]
        L.msclip_col2im.argtypes = [vp, ci, vp] + [ci] * 11 + [vp]
        L.msclip_relu_bwd.argtypes = [vp, vp, vp, vp, ll, vp]
        L.msclip_dwpool_bwd.argtypes = [vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.msclip_dwpool_wgrad.argtypes = [vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.msclip_dw3x3_wgrad.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp]
        L.msclip_bn_stats.argtypes = [vp, ci, ci, vp, ci, ci, ci, vp]
        L.msclip_bn_apply.argtypes = [vp, ci, ci, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, vp]
        L.msclip_bn_bwd_reduce.argtypes = [vp, ci, ci, vp, ci, ci, vp, vp, vp, ci, ci, ci, vp]
        L.msclip_bn_bwd_dx.argtypes = [vp, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, ll, vp]
        L.msclip_bn_finish.argtypes = [vp, ci, ci, ll, vp, vp, cf, vp, vp]
        L.msclip_bn_finish_tiled.argtypes = [vp, ci, ci, ll, vp, vp, cf, vp, ci, vp]
        L.msclip_bn_bwd_finish.argtypes = [vp, ci, ci, ci, vp, vp, vp]
        L.msclip_bn_bwd_fused.argtypes = [ci, vp, ci, vp, ci, vp, ci, ctypes.POINTER(BnBwdSide), ctypes.POINTER(BnBwdSide), ci, ci, ci, ll, vp]
        L.msclip_bn_fold_bwd.argtypes = [vp, ll, vp, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp, vp]
        L.msclip_text_lengths.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, ci, ci, vp]
        L.msclip_embed_tokens_packed.argtypes = [vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp]
        L.msclip_attention_varlen.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
        L.msclip_attention_lastq_varlen.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, ci, vp]
        L.msclip_attention_bwd_varlen.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
        L.msclip_embed_tokens_bwd_packed.argtypes = [vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, vp]
        L.msclip_qkv_attention.argtypes = [ctypes.POINTER(QkvAttnDesc), vp]
        L.msclip_qkvattn_tables.argtypes = [vp, ci, ci, vp, vp, vp, ci, ci, vp]
        L.msclip_pack_weights.argtypes = [vp, vp, ci, ci, vp]
        L.msclip_transpose_bf16_multi.argtypes = [vp, vp, ci, ci, vp]
        L.msclip_colsum_multi.argtypes = [vp, ci, vp]
        L.msclip_image_conv_wgrad.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, vp]
        pp = ctypes.POINTER(vp)
        L.msclip_plan_create.argtypes = [pp]
        L.msclip_plan_destroy.argtypes = [vp]
        L.msclip_plan_begin.argtypes = [vp, pp, ci]
        L.msclip_plan_bind_external.argtypes = [vp, vp, ll]
        L.msclip_plan_record_event.argtypes = [vp, vp]
        L.msclip_plan_wait_event.argtypes = [vp, vp, ci]
        L.msclip_plan_end.argtypes = [vp]
        L.msclip_plan_abort.argtypes = [vp]
        L.msclip_plan_info.argtypes = [vp] + [ctypes.POINTER(ci)] * 5
        L.msclip_plan_op_name.argtypes = [vp, ci]
        L.msclip_plan_op_name.restype = ctypes.c_char_p
        L.msclip_plan_run.argtypes = [vp, pp, ci, pp, ci]
        L.msclip_plan_size.argtypes = [vp]
        L.msclip_plan_probe_enable.argtypes = [vp, ctypes.POINTER(ci), ci, ci]
        L.msclip_plan_probe_disable.argtypes = [vp]
        L.msclip_plan_probe_runs.argtypes = [vp]
        L.msclip_plan_probe_elapsed.argtypes = [vp, ci, ci, ctypes.POINTER(cf)]
        L.msclip_stream_create_cu_masked.argtypes = [ci, ci, pp]
        L.msclip_comm_unique_id.argtypes = [vp]
        L.msclip_comm_init.argtypes = [ci, ci, vp, pp]
        L.msclip_comm_destroy.argtypes = [vp]
        L.msclip_comm_async_error.argtypes = [vp]
        L.msclip_allgather_feats.argtypes = [vp, vp, vp, ll, ci, vp]
        L.msclip_allreduce.argtypes = [vp, vp, vp, ll, ci, ci, vp]
        L.msclip_abi_version.restype = ci
        if L.msclip_abi_version() != ABI_VERSION:          # a stale build of the library (the struct layouts / entry points moved on)
            raise HipUnavailable(f"{LIB_PATH} has ABI version {L.msclip_abi_version()}, this binding needs {ABI_VERSION}: rebuild "
                                 "(bash msclip_amd/csrc/build.sh)")
        L.msclip_build_arch.restype = ctypes.c_char_p
        for name in EXPORTS:
            if name not in ("msclip_build_arch", "msclip_gemm_variant", "msclip_plan_op_name"):
                getattr(L, name).restype = ci
        _lib = L
    return _lib


def env_flag(name):
    """Selection knobs for A/B runs and cross-checks: set to 1 to take the older kernel chain."""
    return os.environ.get(name, "0") == "1"


_PREPARED = set()


def require_gpu():
    if not torch.cuda.is_available():
        raise HipUnavailable("no HIP device visible: msclip_amd has no CPU path (use oracle/ only as a test checker)")
    L = lib()
    dev = torch.cuda.current_device()
    if dev not in _PREPARED:                          # lazy per-device allocations of the library, done outside any capture / plan
        _check(L.msclip_prepare_device(), "msclip_prepare_device")
        _PREPARED.add(dev)


def _check(rc, what):
    if rc != 0:
        raise HipError(f"{what} failed with code {rc} ({'bad arguments' if rc == -1 else 'launch error'})")


def _stream():
    # torch.cuda.current_stream() builds a Stream object per call (~7.6 us: 1 000 launches of a training step = 7.8 ms of host time,
    # on a host that issues the step in 50 ms against the GPU's 58); the raw handle of the calling thread's current stream on its
    # current device is one C call
    return ctypes.c_void_p(_raw_stream(_cur_device()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)
if _raw_stream is None or _cur_device is None or os.environ.get("MSCLIP_STREAM_OBJECT") == "1":   # (another torch build, or the A/B switch: the documented path)
    _raw_stream = lambda _d: torch.cuda.current_stream().cuda_stream
    _cur_device = lambda: 0


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_ZERO = {}


def zero_page(device):
    z = _ZERO.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.bfloat16, device=device)
        _ZERO[device] = z
    return z


def _bf16(t):
    assert t.dtype == torch.bfloat16 and t.is_cuda and t.stride(-1) == 1, (t.dtype, t.device, t.stride())


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous(), (t.dtype, t.device, t.is_contiguous())


_COMPUTE = {}


def off_default_stream(fn):
    """Method decorator (the object carries `.dev`) of the engine's / training step's entry points.  A pass-through unless the
    host application opted in with MSCLIP_AUTO_COMPUTE_STREAM=1: then, while a torch.distributed process group exists, a caller
    that is on the legacy default (null) stream is moved -- ONCE, for good: torch.cuda.set_stream on the calling thread, after
    the new stream has been ordered behind the null stream -- to the per-device compute stream (use_compute_stream).
    Why anyone would want that: with an RCCL communicator in the process every launch on the null stream pays for the legacy
    default stream's implicit synchronisation with the communicator's streams.  ViT-B/32 training step, one-rank RCCL group
    alive (tools/probes/reducer_probe.py): 110 -> 145 ms at batch 512 and 30 -> 55 ms at batch 32 on the null stream, 110 / 33 ms
    on the compute stream.  Switching per call was measured too and is worse than not switching (167 ms), so the move is
    permanent -- which is why it is the HOST's decision: bench.py and tools/train_synthetic.py call hip.use_compute_stream()
    once at start-up (round 5; up to round 4 this decorator moved every caller implicitly)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        if env_flag("MSCLIP_AUTO_COMPUTE_STREAM"):
            import torch.distributed as dist
            dev = self.dev
            if dev.type == "cuda" and dist.is_available() and dist.is_initialized():
                cur = torch.cuda.current_stream(dev)
                if cur == torch.cuda.default_stream(dev) and not torch.cuda.is_current_stream_capturing():
                    use_compute_stream(dev)
        return fn(self, *args, **kwargs)
    return wrapped


def use_compute_stream(device=None):
    """Make the per-device compute stream (highest priority, a hardware queue of its own) the calling thread's current stream,
    ordered behind whatever the previous current stream has queued; returns it.  Call once at start-up in a process that has
    (or will have) an RCCL process group: launches on the legacy default stream synchronise implicitly with the communicator's
    streams (off_default_stream's docstring has the numbers).  Everything the caller issues afterwards on its current stream
    stays ordered with this library's work."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    cs = compute_stream(device)
    cs.wait_stream(torch.cuda.current_stream(device))
    torch.cuda.set_stream(cs)
    return cs


def priority_stream(device, urgent):
    """A non-blocking HIP stream of the HIGHEST (urgent=True) or LOWEST priority on `device`, as a torch stream
    (msclip_stream_create).  The runtime deals hardware queues per priority level, so a stream created here never shares a
    queue with torch's pool streams, RCCL's streams or the null stream (all normal priority), and the command processor
    serves the levels in order.  Never destroyed (a handful per process, cached by the callers)."""
    L = lib()
    with torch.cuda.device(device):
        least, greatest = ctypes.c_int(0), ctypes.c_int(0)
        _check(L.msclip_stream_priority_range(ctypes.byref(least), ctypes.byref(greatest)), "msclip_stream_priority_range")
        ptr = ctypes.c_void_p()
        _check(L.msclip_stream_create(greatest.value if urgent else least.value, ctypes.byref(ptr)), "msclip_stream_create")
    return torch.cuda.ExternalStream(ptr.value, device=device)


def background_stream(device):
    """The stream kind of the weight-gradient lane: a torch pool stream (normal priority).  Lowest
    priority (MSCLIP_LANE_PRIORITY=low) measured the same without collectives but 139-147 ms instead of 110-112 on the
    ViT-B/32 batch-512 step once RCCL collectives are issued in the process (tools/probes/reducer_probe.py)."""
    if os.environ.get("MSCLIP_LANE_PRIORITY", "normal") == "low":
        return priority_stream(device, False)
    return torch.cuda.Stream(device=device)


def compute_stream(device):
    """The per-device compute stream off_default_stream moves null-stream work to (highest priority, a queue of its own)."""
    cs = _COMPUTE.get(device)
    if cs is None:
        cs = _COMPUTE[device] = (torch.cuda.Stream(device=device) if os.environ.get("MSCLIP_COMPUTE_PRIORITY", "high") == "normal"
                                 else priority_stream(device, True))
    return cs


class KernelProbe:
    """Brackets every launch of one kernel family with HIP events on the launch stream (bench.py's roofline leg).
    `units` is the algorithmic work of the launch (FLOPs for the GEMM)."""

    def __init__(self):
        self.records = []          # (start_event, end_event, units)
        self.tags = []             # optional per-launch description
        self.bytes = []            # algorithmic HBM bytes of the launch (operands once + residual once + output once)
        self.rows = []             # (device row count or None, the launch's host-side M): launches sized by a device-side count

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, start, units, tag=None, nbytes=0, rows=None):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.append((start, ev, units))
        self.tags.append(tag)
        self.bytes.append(nbytes)
        self.rows.append(rows)

    def resolve_rows(self):
        """Launches whose row count lives on the device (mdev) were recorded with their upper bound M: scale their FLOPs / bytes /
        shape tags to the rows that really ran (one host read per distinct device counter; call after a synchronize and before
        summary() / by_shape())."""
        cache = {}
        for i, r in enumerate(self.rows):
            if r is None or r[0] is None:
                continue
            t, bound = r
            key = t.data_ptr()
            if key not in cache:
                cache[key] = min(int(t.item()), bound)
            f = cache[key] / bound
            s, e, u = self.records[i]
            self.records[i] = (s, e, u * f)
            self.bytes[i] = self.bytes[i] * f
            if self.tags[i] is not None:
                self.tags[i] = (cache[key],) + tuple(self.tags[i][1:])
            self.rows[i] = None

    def by_shape(self):
        """-> {tag: (launches, total_ms, total_units, total_bytes)}; call after a synchronize."""
        out = {}
        for (s, e, u), tag, nb in zip(self.records, self.tags, self.bytes):
            n, ms, uu, bb = out.get(tag, (0, 0.0, 0.0, 0))
            out[tag] = (n + 1, ms + s.elapsed_time(e), uu + u, bb + nb)
        return out

    def summary(self):
        """-> (launches, total_ms, total_units); call after a synchronize."""
        ms = sum(s.elapsed_time(e) for s, e, _ in self.records)
        return len(self.records), ms, sum(u for _, _, u in self.records)


_REC = [None]        # the hip.Plan that is recording (hip.gemm / gemm_f8 note what their table entries compute)


class Plan:
    """A recorded launch table (include/msclip_hip.h "Launch plans", csrc/plan.h): record one step while it runs, replay it natively.

        plan = hip.Plan(dev, [main, side, ...])      # torch streams: slot i = streams[i]; slot 0 must be current while recording
        with plan.recording(externals=[img, tok]):
            ... the step, issued as usual (it really runs) ...; cross-stream edges through plan.record_event / plan.wait_event
        plan.run([main, side, ...], [img2, tok2])    # any later batch of the same shapes

    The table holds raw device addresses: every buffer the step touches (other than the externals) must stay alive and in place
    for as long as the plan does -- the owner (the engine) drops its plans whenever it re-packs weights or frees a workspace."""

    def __init__(self, device, streams):
        self.dev = device
        self.n_streams = len(streams)
        self._h = ctypes.c_void_p()
        _check(lib().msclip_plan_create(ctypes.byref(self._h)), "msclip_plan_create")
        self._rec_streams = list(streams)
        self._ext_shapes = None
        self.ready = False
        self.n_ops = self.n_launches = self.n_events = 0
        self.keep = []                                    # tensors the table points into that nothing else keeps alive
        self.meta = []                                    # (entry index, kind, flops, tag, bytes, device row counter, bound M)
        self._probe = None

    def note(self, kind, flops, tag, nbytes, mdev, M):
        """Called by the GEMM bindings during the recording: the entry the launch about to be issued will occupy."""
        self.meta.append((lib().msclip_plan_size(self._h), kind, flops, tag, nbytes, mdev, M))

    def enable_probe(self, kinds, runs):
        """HIP timing events around every recorded GEMM entry whose kind ("gemm:pp", "gemm:ppconv", "gemm:stream", "gemm:dense128",
        "gemm_f8", ...) is in `kinds`, for the next `runs` replays; -> number of entries probed."""
        sel = [m for m in self.meta if m[1] in kinds]
        if not sel:
            return 0
        arr = (ctypes.c_int * len(sel))(*[m[0] for m in sel])
        _check(lib().msclip_plan_probe_enable(self._h, arr, len(sel), int(runs)), "msclip_plan_probe_enable")
        self._probe = sel
        return len(sel)

    def probe_results(self, disable=True):
        """-> [(kind, ms, flops, tag, bytes)] per probed launch per run (device row counters resolved: FLOPs / bytes of the rows
        that ran).  Call after a synchronize."""
        sel, out, cache = self._probe or [], [], {}
        runs = lib().msclip_plan_probe_runs(self._h)
        ms = ctypes.c_float()
        for r in range(max(runs, 0)):
            for i, (_, kind, flops, tag, nbytes, mdev, M) in enumerate(sel):
                _check(lib().msclip_plan_probe_elapsed(self._h, r, i, ctypes.byref(ms)), "msclip_plan_probe_elapsed")
                f = 1.0
                if mdev is not None:
                    k = mdev.data_ptr()
                    if k not in cache:
                        cache[k] = min(int(mdev.item()), M)
                    f = cache[k] / M
                    tag = (cache[k],) + tuple(tag[1:])
                out.append((kind, ms.value, flops * f, tag, nbytes * f))
        if disable:
            lib().msclip_plan_probe_disable(self._h)
            self._probe = None
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().msclip_plan_destroy(h)
            except Exception:
                pass

    @staticmethod
    def _handles(streams):
        arr = (ctypes.c_void_p * len(streams))(*[ctypes.c_void_p(s.cuda_stream) for s in streams])
        return arr

    def recording(self, externals=()):
        plan = self

        class _Rec:
            def __enter__(self_):
                arr = plan._handles(plan._rec_streams)
                _check(lib().msclip_plan_begin(plan._h, arr, plan.n_streams), "msclip_plan_begin")
                _REC[0] = plan
                plan._ext_shapes = []
                for t in externals:
                    idx = lib().msclip_plan_bind_external(plan._h, _p(t), t.numel() * t.element_size())
                    if idx < 0:
                        lib().msclip_plan_abort(plan._h)
                        raise HipError("msclip_plan_bind_external failed")
                    plan._ext_shapes.append((tuple(t.shape), t.dtype))
                return plan

            def __exit__(self_, et, ev, tb):
                _REC[0] = None
                if et is not None:
                    lib().msclip_plan_abort(plan._h)
                    return False
                n = lib().msclip_plan_end(plan._h)
                if n <= 0:
                    raise HipError("msclip_plan_end: the recording is unusable (a launch on a stream outside the plan's slots, an "
                                   "entry point with host-array arguments, or nothing was launched)")
                a, b, c, d, e = (ctypes.c_int() for _ in range(5))
                lib().msclip_plan_info(plan._h, *[ctypes.byref(x) for x in (a, b, c, d, e)])
                plan.n_ops, plan.n_launches, plan.n_events = a.value, b.value, c.value
                plan.ready = True
                return False
        return _Rec()

    def record_event(self, stream):
        """-> event id: 'everything queued on `stream` so far' (the caller records its own torch event for the eager pass)."""
        eid = lib().msclip_plan_record_event(self._h, ctypes.c_void_p(stream.cuda_stream))
        if eid < 0:
            raise HipError("msclip_plan_record_event failed (stream not one of the plan's slots?)")
        return eid

    def wait_event(self, stream, eid):
        _check(lib().msclip_plan_wait_event(self._h, ctypes.c_void_p(stream.cuda_stream), eid), "msclip_plan_wait_event")

    def op_names(self):
        return [lib().msclip_plan_op_name(self._h, i).decode() for i in range(self.n_ops)]

    def run(self, streams, externals=()):
        assert self.ready and len(streams) == self.n_streams and len(externals) == len(self._ext_shapes)
        for t, (shape, dtype) in zip(externals, self._ext_shapes):
            if tuple(t.shape) != shape or t.dtype != dtype or not t.is_contiguous():
                raise ValueError(f"plan external: expected a contiguous {dtype} tensor of shape {shape}, got {t.dtype} {tuple(t.shape)}")
        ext = (ctypes.c_void_p * max(len(externals), 1))(*[ctypes.c_void_p(t.data_ptr()) for t in externals])
        _check(lib().msclip_plan_run(self._h, self._handles(streams), self.n_streams, ext, len(externals)), "msclip_plan_run")


class PlanProbeResults:
    """KernelProbe's reading interface (summary / by_shape / bytes) over Plan.probe_results() rows (ms, flops, tag, bytes)."""

    def __init__(self, rows):
        self.rows_ = list(rows)
        self.bytes = [r[3] for r in self.rows_]

    def resolve_rows(self):
        pass

    def summary(self):
        return len(self.rows_), sum(r[0] for r in self.rows_), sum(r[1] for r in self.rows_)

    def by_shape(self):
        out = {}
        for ms, u, tag, nb in self.rows_:
            n, m0, u0, b0 = out.get(tag, (0, 0.0, 0.0, 0))
            out[tag] = (n + 1, m0 + ms, u0 + u, b0 + nb)
        return out


_NCCL_DTYPE = {torch.bfloat16: 0, torch.float32: 1, torch.uint8: 2, torch.int32: 3}


def allgather_feats(comm, send, recv):
    """recv [world * n, ...] = every rank's send [n, ...], rank-major, through the C ABI's RCCL entry point on the CURRENT stream
    (msclip_allgather_feats; reference lib/utils/comm.py:140-154)."""
    assert send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous() and send.dtype == recv.dtype
    assert recv.numel() % send.numel() == 0
    _check(lib().msclip_allgather_feats(comm, _p(send), _p(recv), send.numel(), _NCCL_DTYPE[send.dtype], _stream()), "msclip_allgather_feats")
    return recv


def allreduce(comm, t, op="sum"):
    """In-place element-wise sum / max of `t` over the ranks on the CURRENT stream (msclip_allreduce; lib/utils/utils.py:66-73)."""
    assert t.is_cuda and t.is_contiguous()
    _check(lib().msclip_allreduce(comm, _p(t), _p(t), t.numel(), _NCCL_DTYPE[t.dtype], {"sum": 0, "max": 1}[op], _stream()), "msclip_allreduce")
    return t


def cu_masked_stream(device, n_cus, from_top=True):
    """A torch stream whose kernels are confined to n_cus compute units (msclip_stream_create_cu_masked); never destroyed."""
    ptr = ctypes.c_void_p()
    with torch.cuda.device(device):
        _check(lib().msclip_stream_create_cu_masked(int(n_cus), 1 if from_top else 0, ctypes.byref(ptr)), "msclip_stream_create_cu_masked")
    return torch.cuda.ExternalStream(ptr.value, device=device)


_gemm_probe = {}   # kernel variant -> KernelProbe


def gemm_variant(desc):
    """Which kernel msclip_gemm dispatches this descriptor to -- asked from the library itself (msclip_gemm_variant):
    'pp' = gemm_pp_kernel<0> (dense 256x256 ping-pong, the default for large problems), 'ppconv' = gemm_pp_kernel<1>,
    'stream' = gemm_stream_kernel (K <= 192), 'dense128' = gemm_kernel<0,128,128>, 'conv192'/'conv128' = gemm_kernel<1,...>,
    'w4' = gemm_w4_kernel (tile 7)."""
    return lib().msclip_gemm_variant(ctypes.byref(desc)).decode()


def describe_gemm(mode, M, N, K, tile=0, conv=None, ldx=None, resid_kind=0, rpg=INT_MAX):
    """A shape-only descriptor (dummy non-null pointers) for asking msclip_gemm_variant about a launch without data.
    conv = (H, W, Cin, Ho, Wo, stride, pad) for mode 1."""
    d = GemmDesc()
    d.X = d.W = d.zero = d.out = 1
    d.resid = 1 if resid_kind else None
    d.M, d.N, d.K, d.mode, d.tile = M, N, K, mode, tile
    d.ldx, d.ldw, d.ldo, d.ldr = (ldx if ldx is not None else K), K, N, N
    d.rpg, d.resid_kind, d.alpha = rpg, resid_kind, 1.0
    if mode == 1:
        d.H, d.Wd, d.Cin, d.Ho, d.Wo, d.stride, d.pad = conv
        d.ktab = 1
    return d


def gemm_bn_two_pass_ok(spec_cin, N, K, M, conv=None):
    """Does msclip_gemm take this convolution in its train-mode BatchNorm modes (msclip_gemm_desc.bn_mode: the streaming kernel's
    statistics / normalise epilogues)?  Asked from the library (msclip_gemm_variant on a shape-only descriptor)."""
    d = describe_gemm(0 if conv is None else 1, M, N, K, conv=conv, ldx=spec_cin)
    d.bn_mode, d.part, d.part_rows = 1, 1, 2048
    return N % 8 == 0 and gemm_variant(d) == "stream"


def python_probes_active():
    """True while a KernelProbe is attached to the Python bindings (hip.gemm / hip.gemm_f8): such a call must run the eager launch
    loop (a plan replay never enters the bindings; plan-level probes: Plan.enable_probe)."""
    return any(p is not None for p in _gemm_probe.values()) or _f8_probe[0] is not None


def set_gemm_probe(variant, probe):
    """variant: a gemm_variant() name, or 0 / 1 for every dense / every implicit-conv launch."""
    _gemm_probe[variant] = probe


ACT_NONE, ACT_QUICKGELU, ACT_RELU = 0, 1, 2

_WG_CAP = [0]


def set_wg_cap(n):
    """Upper bound on the persistent workgroups (= CUs) a msclip_gemm / msclip_gemm_f8 launch may take (0: all of them); returns
    the previous value.  The engine's two-chain schedule sets it around each chain so that the other chain's kernels find CUs."""
    prev = _WG_CAP[0]
    _WG_CAP[0] = int(n)
    return prev

RESID_NONE, RESID_F32, RESID_BF16, RESID_TABLE, RESID_GELUGRAD, RESID_RELUMASK, RESID_ACCUM = 0, 1, 2, 3, 4, 5, 6


class FoldIn:
    """Consumer side of the LayerNorm fold for one projection launch (msclip_gemm_desc.rowstat ...): the GEMM's X operand holds
    bf16 (x - center) rows, `w` (the gemm() argument) / bias are the first row segment's gamma-folded weight and bias' = b + W beta,
    csum its column sums; rows >= split take w2 / bias2 / csum2 (the other modality).  rowstat: fp32 [M, 2] = (rstd, mean * rstd)."""

    def __init__(self, rowstat, csum, w2=None, bias2=None, csum2=None, split=0):
        self.rowstat, self.csum, self.w2, self.bias2, self.csum2, self.split = rowstat, csum, w2, bias2, csum2, split


class FoldOut:
    """Producer side (msclip_gemm_desc.xb ...): out_proj / c_proj also write xb = bf16(out - center[m]) and the per-64-column
    partial sums part [M, N / 64, 2] of (out - center) and its square."""

    def __init__(self, xb, center, part, resid2=None, split=0):
        self.xb, self.center, self.part = xb, center, part
        self.resid2, self.split = resid2, split      # rows >= split read their fp32 residual from resid2[m] (absolute row m)


def gemm(x, w, out, *, M=None, bias=None, resid=None, resid_kind=0, act=0, alpha=1.0, conv=None, ktab=None,
         ldx=None, ldo=None, ldr=None, rpg=INT_MAX, radd=0, roff=0, N=None, tile=0, out2=None, fold_in=None, fold_out=None,
         colsum_part=None, mdev=None, bn_stats_part=None, bn_consts=None):
    """out = epilogue(alpha * x @ w^T).  x: bf16 [M, K] (or NHWC activation when conv=(H, W, Cin, Ho, Wo, stride, pad)),
    w: bf16 [N, Kpad]; out: bf16 or fp32 2-D buffer.  Training-step forms (ping-pong kernel): out2 = second bf16 output that
    receives the value before the activation; resid_kind = RESID_GELUGRAD multiplies by QuickGELU'(resid) (resid bf16) and,
    with colsum_part (fp32 [M / 128, N]), also leaves the column sums of every 128 stored rows there.
    mdev: a 1-element int32 device tensor; the kernel runs min(M, mdev[0]) rows (msclip_gemm_desc.M_dev: packed captions whose row
    count never visits the host; M -- x's row count -- is the bound the launch is sized for)."""
    _bf16(w)
    d = GemmDesc()
    d.X, d.W, d.zero, d.out = x.data_ptr(), w.data_ptr(), zero_page(x.device).data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.N = N if N is not None else w.shape[0]
    d.K = w.shape[1]
    d.ldw = w.stride(0)
    if conv is None:
        d.mode = 0
        d.M = M if M is not None else x.shape[0]
        d.ldx = ldx if ldx is not None else x.stride(0)
    else:
        d.mode = 1
        d.H, d.Wd, d.Cin, d.Ho, d.Wo, d.stride, d.pad = conv
        d.M = M
        d.ktab = ktab.data_ptr()
    d.ldo = ldo if ldo is not None else out.stride(0)
    d.ldr = ldr if ldr is not None else (resid.stride(0) if resid is not None and resid.dim() == 2 else d.ldo)
    d.act, d.resid_kind = act, resid_kind
    d.out_kind = 1 if out.dtype == torch.float32 else 0
    d.alpha = alpha
    d.rpg, d.radd, d.roff = rpg, radd, roff
    d.tile = tile
    d.wg_cap = _WG_CAP[0]
    if mdev is not None:
        assert mdev.dtype == torch.int32 and mdev.is_cuda
        d.M_dev = mdev.data_ptr()
        if d.tile == 0:
            d.tile = 4                                # the persistent ping-pong kernel is the one that reads M_dev (auto would pick the
            #                                           128 x 128 kernel for a bound below ~11 k rows)
    if fold_in is not None:
        f = fold_in
        _f32(f.rowstat), _f32(f.csum)
        assert f.rowstat.numel() >= 2 * d.M and f.csum.numel() >= d.N
        d.rowstat, d.csum = f.rowstat.data_ptr(), f.csum.data_ptr()
        if f.w2 is not None:
            _bf16(f.w2)
            assert f.w2.shape == w.shape and f.w2.stride(0) == w.stride(0) and f.csum2.numel() >= d.N and f.bias2.numel() >= d.N
            d.W2, d.bias2, d.csum2, d.seg_split = f.w2.data_ptr(), f.bias2.data_ptr(), f.csum2.data_ptr(), f.split
    if fold_out is not None:
        f = fold_out
        _bf16(f.xb)
        assert f.xb.shape[0] >= d.M and f.center.numel() >= d.M and f.part.numel() >= d.M * (d.N // 64) * 2
        assert f.center.dtype == torch.float32 and f.part.dtype == torch.float32
        d.xb, d.ldxb, d.center, d.part = f.xb.data_ptr(), f.xb.stride(0), f.center.data_ptr(), f.part.data_ptr()
        if f.resid2 is not None:
            _f32(f.resid2)
            assert f.resid2.stride(0) == d.ldr and f.resid2.shape[0] >= d.M and 0 < f.split < d.M and f.split % 256 == 0
            d.resid2, d.seg_split = f.resid2.data_ptr(), f.split
    if out2 is not None:
        assert out2.dtype == torch.bfloat16 and out2.stride(0) == d.ldo and out.dtype == torch.bfloat16
        d.out2 = out2.data_ptr()
    if colsum_part is not None:
        _f32(colsum_part)
        assert fold_out is None and resid_kind == RESID_GELUGRAD and d.M % 128 == 0
        assert colsum_part.is_contiguous() and colsum_part.shape == (d.M // 128, d.N)
        d.part = colsum_part.data_ptr()
    if bn_stats_part is not None:                     # msclip_gemm_desc.bn_mode 1: column sums of the product instead of an output
        _f32(bn_stats_part)
        assert bn_stats_part.is_contiguous() and bn_stats_part.dim() == 2 and bn_stats_part.shape[1] == 2 * d.N and bias is None
        d.bn_mode, d.part, d.part_rows = 1, bn_stats_part.data_ptr(), bn_stats_part.shape[0]
    elif bn_consts is not None:                       # bn_mode 2 (bn_finish's [5][N] rows): out = act(x scale + shift [+ resid]), out2 = xhat
        _f32(bn_consts)
        assert bn_consts.is_contiguous() and tuple(bn_consts.shape) == (5, d.N) and out2 is not None and bias is None
        d.bn_mode, d.bn_consts = 2, bn_consts.data_ptr()
    probe = (_gemm_probe.get(d.mode) or _gemm_probe.get(gemm_variant(d))) if _gemm_probe else None
    rec = _REC[0]
    if probe is not None or rec is not None:
        k_alg = d.K if conv is None else conv[2] * (ktab_taps(ktab) if ktab is not None else 1)
        esz = 4 if d.out_kind else 2
        x_bytes = d.M * d.K * 2 if conv is None else (d.M // (conv[3] * conv[4])) * conv[0] * conv[1] * conv[2] * 2
        r_bytes = {RESID_NONE: 0, RESID_F32: d.M * d.N * 4, RESID_BF16: d.M * d.N * 2, RESID_TABLE: 0, RESID_GELUGRAD: d.M * d.N * 2,
                   RESID_RELUMASK: d.M * d.N * 2, RESID_ACCUM: d.M * d.N * 2}[resid_kind]
        nbytes = x_bytes + d.N * d.K * 2 + d.M * d.N * esz * (2 if out2 is not None else 1) + r_bytes + (d.N * 4 if bias is not None else 0)
        if fold_out is not None:                      # bf16 centred copy + per-64-column partial sums + the rows' centres
            nbytes += d.M * d.N * 2 + d.M * (d.N // 64) * 8 + d.M * 4
        if fold_in is not None:                       # (rstd, mean * rstd) per row, per tile column; second weight
            nbytes += d.M * 8 * ((d.N + 255) // 256) + (d.N * d.K * 2 if fold_in.w2 is not None else 0)
        flops, tag = 2.0 * d.M * d.N * k_alg, (d.M, d.N, d.K, k_alg, conv is not None, act, resid_kind, d.out_kind)
        if rec is not None:                           # a launch plan is recording: what this table entry computes (Plan.enable_probe)
            rec.note("gemm:" + gemm_variant(d), flops, tag, nbytes, mdev, d.M)
        if probe is not None:
            t0 = probe.begin()
            _check(lib().msclip_gemm(ctypes.byref(d), _stream()), "msclip_gemm")
            probe.end(t0, flops, tag, nbytes, rows=(mdev, d.M))
            return out
    _check(lib().msclip_gemm(ctypes.byref(d), _stream()), "msclip_gemm")
    return out


F8 = torch.float8_e4m3fn
_f8_probe = [None]


def set_gemm_f8_probe(probe):
    _f8_probe[0] = probe


def quantize_rows_f8(w):
    """Host-side (pack-time) per-row e4m3 quantisation of an fp32 / bf16 matrix [N, K]: -> (uint8 [N, K], fp32 scale [N])
    with w ~= q * scale[:, None]; scale = max |row| / 448 (the format's largest finite value)."""
    wf = w.detach().float()
    s = wf.abs().amax(dim=1).clamp_min(1e-30) / 448.0
    q = (wf / s[:, None]).to(F8)
    return q.view(torch.uint8).contiguous(), s.contiguous()


def gemm_f8(xq, wq, out, row_scale, col_scale, *, M=None, bias=None, resid=None, resid_kind=0, act=0, alpha=1.0, out_scale=1.0,
            fold_out=None, mdev=None):
    """out = epilogue(alpha * row_scale[m] * col_scale[n] * xq @ wq^T) on the fp8 MX MFMA; xq uint8 [M, K] / wq uint8 [N, K]
    hold OCP e4m3 bytes, K % 128 == 0.  out uint8: an e4m3 output, stored value = fp8(epilogue value * out_scale)."""
    assert xq.dtype == torch.uint8 and wq.dtype == torch.uint8 and xq.stride(-1) == 1 and wq.stride(-1) == 1
    _f32(col_scale)
    assert row_scale.dtype == torch.float32
    d = GemmDesc()
    d.X, d.W, d.zero, d.out = xq.data_ptr(), wq.data_ptr(), zero_page(xq.device).data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.M = M if M is not None else xq.shape[0]
    d.N, d.K = wq.shape[0], wq.shape[1]
    d.ldx, d.ldw, d.ldo = xq.stride(0), wq.stride(0), out.stride(0)
    d.ldr = resid.stride(0) if resid is not None and resid.dim() == 2 else d.ldo
    d.mode, d.act, d.resid_kind, d.alpha, d.rpg = 0, act, resid_kind, alpha, INT_MAX
    d.out_kind = 1 if out.dtype == torch.float32 else 2 if out.dtype == torch.uint8 else 0
    d.out_scale = out_scale
    d.wg_cap = _WG_CAP[0]
    if mdev is not None:
        assert mdev.dtype == torch.int32 and mdev.is_cuda
        d.M_dev = mdev.data_ptr()
    if fold_out is not None:                          # c_proj as the producer of the next block's folded ln_1 (FoldOut)
        f = fold_out
        _bf16(f.xb)
        assert f.xb.shape[0] >= d.M and f.center.numel() >= d.M and f.part.numel() >= d.M * (d.N // 64) * 2
        d.xb, d.ldxb, d.center, d.part = f.xb.data_ptr(), f.xb.stride(0), f.center.data_ptr(), f.part.data_ptr()
    assert row_scale.numel() >= d.M and col_scale.numel() >= d.N
    probe, rec = _f8_probe[0], _REC[0]
    if probe is not None or rec is not None:
        esz = {0: 2, 1: 4, 2: 1}[d.out_kind]
        r_bytes = {RESID_NONE: 0, RESID_F32: d.M * d.N * 4, RESID_BF16: d.M * d.N * 2}[resid_kind]
        nbytes = d.M * d.K + d.N * d.K + d.M * d.N * esz + r_bytes + (d.M + d.N) * 4 + (d.N * 4 if bias is not None else 0)
        flops, tag = 2.0 * d.M * d.N * d.K, (d.M, d.N, d.K, d.K, False, act, resid_kind, d.out_kind)
        if rec is not None:
            rec.note("gemm_f8", flops, tag, nbytes, mdev, d.M)
    t0 = probe.begin() if probe is not None else None
    _check(lib().msclip_gemm_f8(ctypes.byref(d), _p(row_scale), _p(col_scale), _stream()), "msclip_gemm_f8")
    if probe is not None:
        probe.end(t0, flops, tag, nbytes, rows=(mdev, d.M))
    return out


def layernorm_f8(x, gamma, beta, gamma2, beta2, split, q, row_scale, M, eps=1e-12, mdev=None):
    """q[m] = e4m3(LN(x[m]) / s[m]), row_scale[m] = s[m]; (gamma, beta) for m < split, (gamma2, beta2) from there on."""
    assert x.dtype == torch.float32 and x.stride(-1) == 1 and q.dtype == torch.uint8 and q.stride(-1) == 1
    _check(lib().msclip_layernorm_f8(_p(x), x.stride(0), _p(gamma), _p(beta), _p(gamma2), _p(beta2), split, _p(q), q.stride(0),
                                     _p(row_scale), M, x.shape[-1], eps, _p(mdev), _stream()), "msclip_layernorm_f8")
    return q


def quant_f8_rows(x, q, row_scale, M=None):
    _bf16(x)
    M = x.shape[0] if M is None else M
    _check(lib().msclip_quant_f8_rows(_p(x), x.stride(0), _p(q), q.stride(0), _p(row_scale), M, x.shape[1], _stream()),
           "msclip_quant_f8_rows")
    return q


def gemm_splitk(x, w, slices, out=None, tile=0):
    """fp32 [N_x, N_w] = x @ w^T for two K-contiguous bf16 operands x [M, K], w [N, K] with a deep K (K % (64*slices) == 0):
    `slices` K ranges contracted by separate workgroup rows of ONE launch into fp32 partials, folded in a fixed order.
    tile = 0: 128 x 128 tiles (narrow outputs); tile = 4: the 256 x 256 ping-pong kernel (wide outputs: 9-36 tiles x slices
    workgroups fill the chip)."""
    _bf16(x)
    _bf16(w)
    M, N, K = x.shape[0], w.shape[0], x.shape[1]
    assert w.shape[1] == K and K % (64 * slices) == 0
    part = torch.empty(slices, M * N, dtype=torch.float32, device=x.device)
    d = GemmDesc()
    d.X, d.W, d.zero, d.out = x.data_ptr(), w.data_ptr(), zero_page(x.device).data_ptr(), part.data_ptr()
    d.M, d.N, d.K, d.ldx, d.ldw, d.ldo = M, N, K, x.stride(0), w.stride(0), N
    d.out_kind, d.alpha, d.rpg, d.tile = 1, 1.0, INT_MAX, tile
    _check(lib().msclip_gemm_splitk(ctypes.byref(d), slices, _stream()), "msclip_gemm_splitk")
    if slices > 1:
        return colsum(part, out=out.view(-1) if out is not None else None).view(M, N)
    if out is not None:
        out.view(-1).copy_(part.view(-1))
        return out
    return part.view(M, N)


def gemm_splitk_tn(dy, x, T, slices, out=None):
    """fp32 [N_dy, N_x] = dy[:T]^T @ x[:T] for TOKEN-major bf16 operands dy [*, N_dy], x [*, N_x] (what a weight gradient is
    made of) without transposing them: msclip_gemm_splitk_tn, `slices` token ranges into fp32 partials folded in a fixed order.
    Channel counts need not be whole tiles (N_x a multiple of 4, rows 16-byte aligned): an edge tile's extra channels are
    whatever lies to the right of the operand in memory, and land only in outputs that are not stored."""
    _bf16(dy)
    _bf16(x)
    Mo, No = dy.shape[1], x.shape[1]
    assert dy.shape[0] >= T and x.shape[0] >= T and No % 4 == 0 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
    part = torch.empty(slices, Mo * No, dtype=torch.float32, device=dy.device) if (slices > 1 or out is None) else out.view(1, -1)
    d = GemmDesc()
    d.X, d.W, d.zero, d.out = dy.data_ptr(), x.data_ptr(), zero_page(dy.device).data_ptr(), part.data_ptr()
    d.M, d.N, d.K, d.ldx, d.ldw, d.ldo = Mo, No, T, dy.stride(0), x.stride(0), No
    d.out_kind, d.alpha, d.rpg = 1, 1.0, INT_MAX
    _check(lib().msclip_gemm_splitk_tn(ctypes.byref(d), slices, _stream()), "msclip_gemm_splitk_tn")
    if slices > 1:
        return colsum(part, out=out.view(-1) if out is not None else None).view(Mo, No)
    return part.view(Mo, No)


_TAPS = {}


def ktab_taps(ktab):
    """Number of real (non-padding) filter taps described by a chunk table (cached; used by the probe only)."""
    key = ktab.data_ptr()
    if key not in _TAPS:
        t = ktab.cpu()
        valid = t[t >= 0]
        _TAPS[key] = int(torch.unique(valid >> 20).numel())
    return _TAPS[key]


def attention(qkv, out, nsamples, L, heads, causal):
    """qkv: bf16 [nsamples*L, 3*heads*64] (a row-range view is fine); out: bf16 [nsamples*L, heads*64]."""
    _bf16(qkv)
    _bf16(out)
    assert qkv.shape[0] == nsamples * L and out.shape[0] == nsamples * L
    _check(lib().msclip_attention(_p(qkv), _p(out), nsamples, L, heads, qkv.stride(0), out.stride(0), int(causal),
                                  _stream()), "msclip_attention")
    return out


def attention_lastq(q, qkv, out, nsamples, L, heads, *, last_row=None, row_base=0):
    """One query per sample (the last block's class / EOT rows).  q: bf16 [nsamples, heads*64]; qkv: the token matrix of
    attention() (its q columns are not read); sample b's keys are the rows row_base + b*L ... up to last_row[b] (int32
    absolute row numbers) or all L of them.  out: bf16 [nsamples, heads*64]."""
    _bf16(q)
    _bf16(qkv)
    _bf16(out)
    assert q.shape[0] >= nsamples and out.shape[0] >= nsamples and qkv.shape[0] >= row_base + nsamples * L
    assert qkv.shape[1] == 3 * heads * 64
    if last_row is not None:
        assert last_row.dtype == torch.int32 and last_row.numel() >= nsamples
    _check(lib().msclip_attention_lastq(_p(q), q.stride(0), _p(qkv), qkv.stride(0), _p(out), out.stride(0), nsamples, L, heads,
                                        _p(last_row) if last_row is not None else None, row_base, _stream()),
           "msclip_attention_lastq")
    return out


def layernorm(x, gamma, beta, out, M, *, row_idx=None, row_mul=1, row_add=0, eps=1e-12, raw_out=None):
    """out[m] = LN(x[row_idx[m] or m*row_mul + row_add]); x fp32 2-D (a row-range view is fine)."""
    assert x.dtype == torch.float32 and x.is_cuda and x.stride(-1) == 1
    C = x.shape[-1]
    _check(lib().msclip_layernorm(_p(x), x.stride(0), _p(row_idx), row_mul, row_add, _p(gamma), _p(beta), _p(out),
                                  out.stride(0), 1 if out.dtype == torch.float32 else 0, _p(raw_out),
                                  raw_out.stride(0) if raw_out is not None else 0, M, C, eps, _stream()),
           "msclip_layernorm")
    return out


def layernorm_stats(x, gamma, beta, out, M, center, rowstat, *, eps=1e-12, raw_out=None, mdev=None):
    """out[m] = LN(x[m]) over contiguous rows, plus the LayerNorm fold's per-row state: center[m] = mean of the row,
    rowstat[m] = (1, 0) (the consuming projection takes `out` as it is)."""
    assert x.dtype == torch.float32 and x.is_cuda and x.stride(-1) == 1
    _check(lib().msclip_layernorm_stats(_p(x), x.stride(0), _p(gamma), _p(beta), _p(out), out.stride(0),
                                        1 if out.dtype == torch.float32 else 0, _p(raw_out),
                                        raw_out.stride(0) if raw_out is not None else 0, _p(center), _p(rowstat), M, x.shape[-1],
                                        eps, _p(mdev), _stream()), "msclip_layernorm_stats")
    return out


def rowstat_finalize(part, center, rowstat, M, C, eps=1e-12, mdev=None):
    """rowstat[m] = (rstd, mu * rstd) and center[m] += mu from a producing GEMM's partial sums part [M, C / 64, 2]."""
    _check(lib().msclip_rowstat_finalize(_p(part), C // 64, _p(center), _p(rowstat), M, C, eps, _p(mdev), _stream()), "msclip_rowstat_finalize")


def layernorm_split(x, gamma, beta, gamma2, beta2, split, out, M, eps=1e-12):
    """out[m] = LN(x[m]) with (gamma, beta) for m < split and (gamma2, beta2) from there on."""
    assert x.dtype == torch.float32 and x.is_cuda and x.stride(-1) == 1
    _check(lib().msclip_layernorm_split(_p(x), x.stride(0), _p(gamma), _p(beta), _p(gamma2), _p(beta2), split, _p(out),
                                        out.stride(0), 1 if out.dtype == torch.float32 else 0, M, x.shape[-1], eps,
                                        _stream()), "msclip_layernorm_split")
    return out


def embed_tokens(tokens, emb, pos, x, eot_row, row_base):
    B, L = tokens.shape
    assert tokens.dtype == torch.int64 and tokens.is_contiguous()
    _check(lib().msclip_embed_tokens(_p(tokens), _p(emb), _p(pos), _p(x), x.stride(0), _p(eot_row), B, L, x.shape[1],
                                     emb.shape[0], row_base, _stream()), "msclip_embed_tokens")


def text_lengths(tokens, length, cu, eot_row=None, row_base=0, dims=None, pad_to=0, cap_rows=0):
    """Packed captions: length[b] = argmax_l tokens[b, l] + 1 (the rows that can influence the EOT row under the causal mask),
    cu[:B] = exclusive prefix sums, cu[B] = total, cu[B + 1] = max length; eot_row[b] = row_base + cu[b] + length[b] - 1.
    dims (int32 [8], optional): the device-side row counts (total, longest, total rounded up to pad_to, row_base + that, padding
    rows, row_base + total, row_base) that launches read through their mdev / dims arguments instead of a host-side size."""
    B, L = tokens.shape
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and tokens.is_cuda
    assert length.dtype == torch.int32 and length.numel() >= B and cu.dtype == torch.int32 and cu.numel() >= B + 2
    assert eot_row is None or (eot_row.dtype == torch.int32 and eot_row.numel() >= B)
    if dims is not None:
        assert dims.dtype == torch.int32 and dims.numel() >= 8 and cap_rows >= B * L
    _check(lib().msclip_text_lengths(_p(tokens), B, L, row_base, _p(length), _p(cu), _p(eot_row), _p(dims), pad_to, cap_rows, _stream()),
           "msclip_text_lengths")


def embed_tokens_packed(tokens, emb, pos, x, cu, row_base, rows_padded, rows_dev=None):
    """x[row_base + cu[b] + l] = emb[tokens[b, l]] + pos[l] for the live positions; rows up to row_base + rows_padded zeroed."""
    B, L = tokens.shape
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and cu.dtype == torch.int32 and cu.numel() >= B + 2
    assert x.shape[0] >= row_base + rows_padded
    _check(lib().msclip_embed_tokens_packed(_p(tokens), _p(emb), _p(pos), _p(x), x.stride(0), _p(cu), B, L, x.shape[1], emb.shape[0],
                                            row_base, rows_padded, _p(rows_dev), _stream()), "msclip_embed_tokens_packed")


def attention_varlen(qkv, out, cu, nsamples, Lmax, heads, causal, pad_rows=0, dims=None):
    """attention() over packed captions: sample b = rows cu[b] .. cu[b + 1] of qkv / out (views that start at the text
    segment); the pad_rows output rows behind cu[nsamples] are zeroed."""
    _bf16(qkv)
    _bf16(out)
    assert cu.dtype == torch.int32 and cu.numel() >= nsamples + 2 and 0 <= pad_rows < 256
    _check(lib().msclip_attention_varlen(_p(qkv), _p(out), _p(cu), nsamples, Lmax, heads, qkv.stride(0), out.stride(0), int(causal),
                                         pad_rows, _p(dims), _stream()), "msclip_attention_varlen")
    return out


def attention_lastq_varlen(q, qkv, out, nsamples, Lmax, heads, cu, row_base=0):
    """attention_lastq over packed captions: sample b's keys are rows row_base + cu[b] .. row_base + cu[b + 1] of qkv."""
    _bf16(q)
    _bf16(qkv)
    _bf16(out)
    assert q.shape[0] >= nsamples and out.shape[0] >= nsamples and qkv.shape[1] == 3 * heads * 64
    assert cu.dtype == torch.int32 and cu.numel() >= nsamples + 2
    _check(lib().msclip_attention_lastq_varlen(_p(q), q.stride(0), _p(qkv), qkv.stride(0), _p(out), out.stride(0), nsamples, Lmax,
                                               heads, _p(cu), row_base, _stream()), "msclip_attention_lastq_varlen")
    return out


def head_major_qkv(w, b, heads, *extra):
    """Packed in_proj rows [q (heads*64) | k | v] -> head-major [q_h | k_h | v_h] per head: the row order msclip_qkv_attention
    expects.  w [3*heads*64, K], b (and every tensor in `extra`, e.g. the fold's column sums) [3*heads*64]."""
    D = heads * 64
    idx = torch.arange(3 * D, device=w.device).view(3, heads, 64).permute(1, 0, 2).reshape(-1)
    return (w[idx].contiguous(), b[idx].contiguous()) + tuple(t[idx].contiguous() for t in extra)


class QkvAttnTables:
    """Row / tile tables of msclip_qkv_attention for one batch layout: cu int32 [nsamples + 1] = first row of every sample
    (device); split_sample = first sample of the second modality (0: none)."""

    def __init__(self, cu, nsamples, split_sample=0, total_rows=None, max_rows=256):
        dev = cu.device
        assert cu.dtype == torch.int32 and cu.numel() >= nsamples + 1
        rows = int(total_rows) if total_rows is not None else int(cu[nsamples])
        self.cu, self.nsamples, self.max_rows = cu, nsamples, max_rows
        self.max_tiles = nsamples + 1
        self.rowseg = torch.zeros(max(rows, 1), 2, dtype=torch.int32, device=dev)
        self.tile_first = torch.zeros(self.max_tiles + 1, dtype=torch.int32, device=dev)
        self.ntiles = torch.zeros(1, dtype=torch.int32, device=dev)
        _check(lib().msclip_qkvattn_tables(_p(cu), nsamples, split_sample, _p(self.rowseg), _p(self.tile_first), _p(self.ntiles),
                                           self.max_tiles, max_rows, _stream()), "msclip_qkvattn_tables")


def qkv_attention(x, w_hm, b_hm, out, tables, heads, *, causal_from_row=INT_MAX, fold_in=None, M=None):
    """out = attention(in_proj(x)) in one launch (msclip_qkv_attention).  w_hm / b_hm: head-major weight / bias (head_major_qkv);
    fold_in: hip.FoldIn whose csum / W2 / bias2 / csum2 are head-major too."""
    _bf16(x); _bf16(out); _bf16(w_hm); _f32(b_hm)
    d = QkvAttnDesc()
    d.X, d.W, d.zero, d.out, d.bias = x.data_ptr(), w_hm.data_ptr(), zero_page(x.device).data_ptr(), out.data_ptr(), b_hm.data_ptr()
    d.cu, d.tile_first, d.rowseg, d.ntiles_dev = tables.cu.data_ptr(), tables.tile_first.data_ptr(), tables.rowseg.data_ptr(), tables.ntiles.data_ptr()
    d.M = M if M is not None else x.shape[0]
    d.K, d.heads, d.ntiles = w_hm.shape[1], heads, 0
    d.ldx, d.ldw, d.ldo = x.stride(0), w_hm.stride(0), out.stride(0)
    d.causal_from_row = causal_from_row
    assert w_hm.shape[0] == heads * 192
    if fold_in is not None:
        f = fold_in
        d.rowstat, d.csum = f.rowstat.data_ptr(), f.csum.data_ptr()
        if f.w2 is not None:
            d.W2, d.bias2, d.csum2, d.seg_split = f.w2.data_ptr(), f.bias2.data_ptr(), f.csum2.data_ptr(), f.split
    _check(lib().msclip_qkv_attention(ctypes.byref(d), _stream()), "msclip_qkv_attention")
    return out


def fill_cls(cls, pos, x, B, L):
    _check(lib().msclip_fill_cls(_p(cls), _p(pos), _p(x), x.stride(0), B, L, x.shape[1], _stream()), "msclip_fill_cls")


def adapter_combine_ln(xin, t, dww, dwb, gamma, beta, xout, B, L, g, usecls, eps=1e-12):
    _check(lib().msclip_adapter_combine_ln(_p(xin), xin.stride(0), _p(t), t.stride(0), _p(dww), _p(dwb), _p(gamma),
                                           _p(beta), _p(xout), xout.stride(0), B, L, g, xin.shape[1], int(usecls), eps,
                                           _stream()), "msclip_adapter_combine_ln")


def adapter_combine_ln_stats(xin, t, dww, dwb, gamma, beta, xout, gamma1, beta1, lno, center, rowstat, B, L, g, usecls, eps=1e-12):
    """adapter_combine_ln plus the block's ln_1 of each row from registers: lno = bf16 LN(xout; gamma1, beta1), center = row
    means of xout, rowstat = (1, 0) (what layernorm_stats(xout, ...) would leave)."""
    _bf16(lno)
    assert center.dtype == torch.float32 and rowstat.dtype == torch.float32 and center.numel() >= B * L and rowstat.numel() >= 2 * B * L
    _check(lib().msclip_adapter_combine_ln_stats(_p(xin), xin.stride(0), _p(t), t.stride(0), _p(dww), _p(dwb), _p(gamma), _p(beta),
                                                 _p(xout), xout.stride(0), _p(gamma1), _p(beta1), _p(lno), lno.stride(0),
                                                 _p(center), _p(rowstat), B, L, g, xin.shape[1], int(usecls), eps, _stream()),
           "msclip_adapter_combine_ln_stats")


def patchify(img, out, B, S, P, kpad):
    """out[b * g * g + py * g + px][c * P * P + kh * P + kw] = bf16(img[b, c, py * P + kh, px * P + kw]), zero-padded to kpad
    columns: the patch matrix of a kernel == stride == P convolution (M.py:2502-2508, 2657).  img NCHW fp32 / bf16 [B, 3, S, S]."""
    assert img.is_cuda and img.is_contiguous() and img.dtype in (torch.float32, torch.bfloat16) and tuple(img.shape) == (B, 3, S, S)
    _bf16(out)
    assert out.shape[0] >= B * (S // P) ** 2 and out.shape[1] == kpad and out.stride(0) == kpad
    _check(lib().msclip_patchify(_p(img), int(img.dtype == torch.bfloat16), _p(out), kpad, B, S, S, P, _stream()), "msclip_patchify")
    return out


def l2norm(x, out_f32=None, out_bf16=None):
    M, E = x.shape
    _check(lib().msclip_l2norm(_p(x), x.stride(0), _p(out_f32), out_f32.stride(0) if out_f32 is not None else 0,
                               _p(out_bf16), out_bf16.stride(0) if out_bf16 is not None else 0, M, E, _stream()),
           "msclip_l2norm")


def gather_rows(x, out, M, *, row_idx=None, row_mul=1, row_add=0):
    """out[m] = x[row_idx[m] or m*row_mul + row_add] for 2-D tensors of one dtype (rows moved as 16-byte pieces)."""
    assert x.dtype == out.dtype and x.stride(-1) == 1 and out.stride(-1) == 1 and x.shape[1] == out.shape[1]
    es = x.element_size()
    _check(lib().msclip_gather_rows(_p(x), x.stride(0) * es, _p(row_idx), row_mul, row_add, _p(out), out.stride(0) * es, M,
                                    x.shape[1] * es, _stream()), "msclip_gather_rows")
    return out


def stem_conv_dual(img, w, bias, out_a, out_b):
    B, _, H, W = img.shape
    assert img.is_contiguous() and img.dtype in (torch.float32, torch.bfloat16)
    _check(lib().msclip_stem_conv3x3s2_dual(_p(img), int(img.dtype == torch.bfloat16), _p(w), _p(bias), _p(out_a),
                                            _p(out_b), B, H, W, w.shape[1] // 2, _stream()), "msclip_stem_conv3x3s2_dual")


def stem_conv_dual_raw(img, w, out_a, out_b):
    """Both Cin = 3 convolutions' RAW outputs (no BatchNorm, bias or ReLU), fp32 [B * Ho * Wo, 48] each, from one pass over the
    image; w fp32 [27, 96] (row ci * 9 + kh * 3 + kw; conv1's channels, then parallel stage 0's)."""
    B, _, H, W = img.shape
    assert img.is_contiguous() and img.dtype in (torch.float32, torch.bfloat16) and H % 2 == 0 and W % 2 == 0
    _f32(w); _f32(out_a); _f32(out_b)
    assert tuple(w.shape) == (27, 96) and out_a.shape[1] == 48 and out_b.shape[1] == 48
    assert out_a.shape[0] >= B * (H // 2) * (W // 2) and out_b.shape[0] >= B * (H // 2) * (W // 2)
    _check(lib().msclip_stem_conv3x3s2_dual_raw(_p(img), int(img.dtype == torch.bfloat16), _p(w), _p(out_a), _p(out_b), B, H, W,
                                                _stream()), "msclip_stem_conv3x3s2_dual_raw")


def stem_conv_dual_bn(img, w, affine, y_a, y_b, xhat_a, xhat_b, eps=1e-5):
    """Train-mode BatchNorm over both Cin = 3 convolutions in two passes over the image, the raw maps never written
    (msclip_stem_conv3x3s2_dual_stats / _norm): affine = [(gamma, beta)] per convolution; y = relu(BN(conv)) and xhat (the
    normalised value, what the backward reads) bf16 [B * Ho * Wo, 48] each.
    -> [(mean, biased var, rstd, scale, shift)] per convolution (fp32 [48])."""
    B, _, H, W = img.shape
    pix = B * (H // 2) * (W // 2)
    assert img.is_contiguous() and img.dtype in (torch.float32, torch.bfloat16) and H % 2 == 0 and W % 2 == 0
    _f32(w)
    assert tuple(w.shape) == (27, 96) and len(affine) == 2
    for t in (y_a, y_b, xhat_a, xhat_b):
        _bf16(t)
        assert t.is_contiguous() and t.shape[1] == 48 and t.shape[0] >= pix
    dev = img.device
    waves = 4 * min(768, (pix + 127) // 128)
    part = torch.empty(waves, 192, dtype=torch.float32, device=dev)
    st = _stream()
    _check(lib().msclip_stem_conv3x3s2_dual_stats(_p(img), int(img.dtype == torch.bfloat16), _p(w), _p(part), waves, B, H, W, st),
           "msclip_stem_conv3x3s2_dual_stats")
    sums = colsum(part)                                      # [conv][2][48]
    o = torch.empty(2, 5, 48, dtype=torch.float32, device=dev)
    for k, (gamma, beta) in enumerate(affine):
        bn_finish(sums[k * 96:], 48, pix, gamma, beta, eps, o[k])
    _check(lib().msclip_stem_conv3x3s2_dual_norm(_p(img), int(img.dtype == torch.bfloat16), _p(w), _p(o[0]), _p(o[1]), _p(y_a), _p(y_b),
                                                 _p(xhat_a), _p(xhat_b), B, H, W, st), "msclip_stem_conv3x3s2_dual_norm")
    return [tuple(o[k, j] for j in range(5)) for k in range(2)]


def stem_dual_conv3x3s2(img, w, bias, out_b, w2, b2, out2):
    """Both Cin=3 convs + the 3x3/s2 conv that consumes the first one's map (which never reaches HBM)."""
    B, _, H, W = img.shape
    assert img.is_contiguous() and img.dtype in (torch.float32, torch.bfloat16)
    assert w.shape[1] == 96 and w2.dtype == torch.bfloat16 and tuple(w2.shape[1:]) == (448,)
    _check(lib().msclip_stem_dual_conv3x3s2(_p(img), int(img.dtype == torch.bfloat16), _p(w), _p(bias), _p(out_b),
                                            _p(w2), _p(b2), _p(out2), B, H, W, w2.shape[0], _stream()),
           "msclip_stem_dual_conv3x3s2")


def conv1x1_conv3x3s2(x, w1, b1, w2, b2, out, B, H, W):
    """relu(conv3x3/s2(relu(conv1x1(x)))) on NHWC bf16 with 48 channels in and between."""
    _bf16(x)
    assert tuple(w1.shape) == (48, 64) and w1.dtype == torch.bfloat16
    assert w2.dtype == torch.bfloat16 and tuple(w2.shape[1:]) == (448,)
    _check(lib().msclip_conv1x1_conv3x3s2(_p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(out), B, H, W, w2.shape[0],
                                          _stream()), "msclip_conv1x1_conv3x3s2")


def convresblock48_s2(x, w1, b1, w2, b2, w3, wr, b3r, out, B, H, W):
    """Whole stride-2 bottleneck 48 -> 48 -> 96 with projection shortcut (one launch, no intermediate in HBM)."""
    _bf16(x)
    for w, shape in ((w1, (48, 64)), (w2, (48, 448)), (w3, (96, 64)), (wr, (96, 64))):
        assert tuple(w.shape) == shape and w.dtype == torch.bfloat16 and w.is_contiguous(), (tuple(w.shape), shape)
    assert b3r.numel() == 96 and b3r.dtype == torch.float32
    _check(lib().msclip_convresblock48_s2(_p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(wr), _p(b3r), _p(out),
                                          B, H, W, _stream()), "msclip_convresblock48_s2")


def dwpool(top, w, out, B, H, W, C, k):
    _check(lib().msclip_dwpool(_p(top), _p(w), _p(out), out.stride(0), B, H, W, C, k, _stream()), "msclip_dwpool")


def lse_rows(logits, lse):
    R, N = logits.shape
    _check(lib().msclip_lse_rows(_p(logits), logits.stride(0), _p(lse), R, N, _stream()), "msclip_lse_rows")


def clip_loss_partial(lse_img, lse_txt, img_rows, label_off, scale, out):
    R = img_rows.shape[0]
    _check(lib().msclip_clip_loss_partial(_p(lse_img), _p(lse_txt), _p(img_rows), img_rows.stride(0), label_off, R,
                                          scale, _p(out), _stream()), "msclip_clip_loss_partial")


def clip_lse_fused(a, b, scale, label_off, nsplit, part_max, part_sum, diag):
    """Partials of logsumexp_n(scale * a @ b^T) per row of a (bf16 [R, E] vs bf16 [N, E]) + the label logits."""
    _bf16(a)
    _bf16(b)
    R, E = a.shape
    _check(lib().msclip_clip_lse_fused(_p(a), a.stride(0), _p(b), b.stride(0), R, b.shape[0], E, scale, label_off,
                                       nsplit, _p(part_max), _p(part_sum), _p(diag), _stream()), "msclip_clip_lse_fused")


def clip_loss_from_partials(pm_i, ps_i, pm_t, ps_t, diag, scale, out, lse_out=None):
    R, nsplit = pm_i.shape
    _check(lib().msclip_clip_loss_from_partials(_p(pm_i), _p(ps_i), _p(pm_t), _p(ps_t), _p(diag), R, nsplit, scale,
                                                _p(out), _p(lse_out), _stream()), "msclip_clip_loss_from_partials")


# ---------------------------------------------------------------------------------------------------------------------
# backward-pass entry points (csrc/backward.hip, csrc/attention_bwd.hip)
# ---------------------------------------------------------------------------------------------------------------------

def transpose_bf16(x, M=None, Mpad=None):
    """x bf16 [M, C] (row-range view fine) -> new bf16 [C, Mpad] with columns >= M zeroed (Mpad: M rounded up to 64)."""
    _bf16(x)
    M = x.shape[0] if M is None else M
    C = x.shape[1]
    Mpad = (M + 63) // 64 * 64 if Mpad is None else Mpad
    out = torch.empty(C, Mpad, dtype=torch.bfloat16, device=x.device)
    _check(lib().msclip_transpose_bf16(_p(x), x.stride(0), _p(out), Mpad, M, C, Mpad, _stream()), "msclip_transpose_bf16")
    return out


def cast_bf16(x, out=None):
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    M, C = x.shape
    if out is None:
        out = torch.empty(M, C, dtype=torch.bfloat16, device=x.device)
    _check(lib().msclip_cast_bf16(_p(x), x.stride(0), _p(out), out.stride(0), M, C, _stream()), "msclip_cast_bf16")
    return out


def cast_bf16_colsum(x, out=None, fold=True, skip_group=0):
    """-> (bf16 copy of the fp32 matrix x, its column sums fp32 [C]) from one pass over x (fold=False: the per-block partial sums
    [blocks, C] instead, for a FoldPlan)."""
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    M, C = x.shape
    if skip_group:                                       # x: skip_group + 1 rows per sample, the first one (class token) skipped
        assert M % (skip_group + 1) == 0
        M = M // (skip_group + 1) * skip_group
    if out is None:
        out = torch.empty(M, C, dtype=torch.bfloat16, device=x.device)
    blocks = max(1, min(1024, M // 16))
    part = torch.empty(blocks, C, dtype=torch.float32, device=x.device)
    _check(lib().msclip_cast_bf16_colsum(_p(x), x.stride(0), _p(out), out.stride(0), M, C, _p(part), blocks, skip_group, _stream()),
           "msclip_cast_bf16_colsum")
    return out, (colsum(part) if fold else part)


class FoldItem(ctypes.Structure):
    """msclip_fold_item (include/msclip_hip.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("M", ctypes.c_int), ("N", ctypes.c_int), ("ld", ctypes.c_int),
                ("scale_n", ctypes.c_int), ("scale", ctypes.c_float)]


class FoldPlan:
    """Deferred column sums of one backward pass: producers leave per-block partial matrices (fp32 [M, N]), add() books their fold,
    run() folds ALL of them with msclip_colsum_multi (one launch per 96) into one fresh result buffer and hands every caller its
    slice.  Nothing on the backward's critical path reads these sums (LayerNorm parameter gradients, bias gradients): ~130 short
    msclip_colsum launches per training step become two."""

    def __init__(self, device):
        self.device, self.items, self.then, self.keep, self.total = device, [], [], [], 0

    def add(self, part, then, scale_n=0, scale=1.0):
        """part: fp32 [M, N] (row stride >= N); then(result fp32 [N]) is called by run() once the fold is queued."""
        _f32_2d = part.dtype == torch.float32 and part.dim() == 2 and part.stride(1) == 1
        assert _f32_2d and part.is_cuda
        M, N = part.shape
        it = FoldItem()
        it.src, it.M, it.N, it.ld, it.scale_n, it.scale = part.data_ptr(), M, N, part.stride(0), scale_n, scale
        self.items.append((it, self.total))
        self.then.append((self.total, N, then))
        self.keep.append(part)
        self.total += (N + 3) // 4 * 4

    def run(self):
        if not self.items:
            return
        out = torch.empty(self.total, dtype=torch.float32, device=self.device)
        base = out.data_ptr()
        arr = (FoldItem * len(self.items))()
        for i, (it, off) in enumerate(self.items):
            it.dst = base + 4 * off
            arr[i] = it
        _check(lib().msclip_colsum_multi(ctypes.cast(arr, ctypes.c_void_p), len(self.items), _stream()), "msclip_colsum_multi")
        cur = torch.cuda.current_stream(self.device)
        for p in self.keep:
            p.record_stream(cur)                      # (a partial produced on another stream's allocation pool is read here)
        for off, N, then in self.then:
            then(out[off:off + N])
        self.items, self.then, self.keep, self.total = [], [], [], 0


def colsum(x, out=None, M=None, accumulate=False):
    M = x.shape[0] if M is None else M
    N = x.shape[1]
    assert x.dtype in (torch.float32, torch.bfloat16) and x.stride(-1) == 1
    if N <= 384 and M >= 65536 and x.stride(0) == N and x.dtype == torch.bfloat16 and not (N % 8):
        # narrow, long, contiguous (the convolutions' bias sums: 48-384 channels over up to 6.4 M pixels): r consecutive rows
        # are read as ONE row of r*N columns, so every lane works on full cache lines (1.6 -> 4+ TB/s), and the r partial
        # vectors are added at the end
        r = 1
        while r * 2 * N <= 1536 and M % (r * 2) == 0:
            r *= 2
        if r > 1:
            wide = colsum(x[:M].view(M // r, r * N)).view(r, N).sum(0)
            if out is None:
                return wide
            if accumulate:
                out += wide
            else:
                out.copy_(wide)
            return out
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    if M > 2048:
        chunks = min(256, (M + 511) // 512)                           # long matrices: row chunks in parallel, then folded
    else:
        # 64-2048 rows (per-block partials of the LayerNorm backward, of the cast + column-sum pass, of the BatchNorm kernels):
        # one chunk would be (N / 64) workgroups -- 12 for 768 columns -- each walking every row: 26 us for 3 MB.  Enough
        # chunks for ~384 workgroups of >= 32 rows
        chunks = max(1, min(M // 32, -(-384 // ((N + 63) // 64))))
    scratch = torch.empty(chunks, N, dtype=torch.float32, device=x.device) if chunks > 1 else None
    _check(lib().msclip_colsum(_p(x), x.stride(0), int(x.dtype == torch.float32), _p(out), M, N, int(accumulate),
                               _p(scratch), chunks, _stream()), "msclip_colsum")
    return out


def quickgelu(h, y):
    _bf16(h); _bf16(y)
    assert h.is_contiguous() and y.is_contiguous()
    _check(lib().msclip_quickgelu(_p(h), _p(y), h.numel(), _stream()), "msclip_quickgelu")
    return y


def quickgelu_bwd(h, dy, dh):
    assert h.is_contiguous() and dy.is_contiguous() and dh.is_contiguous()
    _check(lib().msclip_quickgelu_bwd(_p(h), _p(dy), _p(dh), h.numel(), _stream()), "msclip_quickgelu_bwd")
    return dh


LN_PART_BLOCKS = 1024


def layernorm_bwd(x, dy, gamma, dx, M, *, row_idx=None, row_mul=1, accumulate=True, want_param_grads=True, eps=1e-12,
                  dxb=None, sum_part=None, sum_accumulate=False, fold=True):
    """-> (dgamma, dbeta) fp32 [C] (or None).  x fp32 [*, C]; dy [M, C] bf16 / fp32; dx fp32 gets (+=) the input gradient
    at the rows the forward read.  dxb (bf16 [M, C]) + sum_part (fp32 [LN_PART_BLOCKS, C]): the written dx rows also as bf16,
    their per-block column sums into (sum_accumulate: onto) sum_part -- colsum(sum_part) = the column sums of the new dx rows."""
    C = x.shape[-1]
    part = torch.empty(LN_PART_BLOCKS, 2, C, dtype=torch.float32, device=x.device) if want_param_grads else None
    if dxb is not None:
        _bf16(dxb); _f32(sum_part)
        assert dxb.shape[0] >= M and sum_part.shape == (LN_PART_BLOCKS, C) and row_idx is None and row_mul == 1
    _check(lib().msclip_layernorm_bwd(_p(x), x.stride(0), _p(row_idx), row_mul, _p(dy), dy.stride(0),
                                      int(dy.dtype == torch.float32), _p(gamma), _p(dx), dx.stride(0), int(accumulate),
                                      _p(part), LN_PART_BLOCKS, M, C, eps, _p(dxb), dxb.stride(0) if dxb is not None else 0,
                                      _p(sum_part), int(sum_accumulate), _stream()), "msclip_layernorm_bwd")
    if not want_param_grads:
        return None, None
    if not fold:                                         # the caller folds (FoldPlan): the partials [LN_PART_BLOCKS, 2 C] = (dgamma | dbeta)
        return part.view(LN_PART_BLOCKS, 2 * C), None
    both = colsum(part.view(LN_PART_BLOCKS, 2 * C))
    return both[:C], both[C:]


def _colsum_part(part, nsamples, heads):
    if part is not None:
        _f32(part)
        assert part.shape == (nsamples, 3 * heads * 64)
    return _p(part)


def attention_bwd(qkv, o, dout, dqkv, nsamples, L, heads, causal, colsum_part=None):
    """colsum_part (fp32 [nsamples, 3 D], L <= 96): also the per-sample token sums of dqkv (the in_proj bias gradient's partials)."""
    _bf16(qkv); _bf16(o); _bf16(dout); _bf16(dqkv)
    assert o.stride(0) == dout.stride(0) and qkv.stride(0) == dqkv.stride(0)
    _check(lib().msclip_attention_bwd(_p(qkv), _p(o), _p(dout), _p(dqkv), nsamples, L, heads, qkv.stride(0), o.stride(0),
                                      int(causal), _colsum_part(colsum_part, nsamples, heads), _stream()), "msclip_attention_bwd")
    return dqkv


def attention_bwd_varlen(qkv, o, dout, dqkv, cu, nsamples, Lmax, heads, causal, pad_rows=0, colsum_part=None):
    """attention_bwd over packed captions (views that start at the text segment); dqkv's pad_rows rows behind cu[nsamples] zeroed."""
    for t in (qkv, o, dout, dqkv):
        _bf16(t)
    assert cu.dtype == torch.int32 and cu.numel() >= nsamples + 2 and 0 <= pad_rows < 256
    _check(lib().msclip_attention_bwd_varlen(_p(qkv), _p(o), _p(dout), _p(dqkv), _p(cu), nsamples, Lmax, heads, qkv.stride(0),
                                             o.stride(0), int(causal), pad_rows, _colsum_part(colsum_part, nsamples, heads),
                                             _stream()), "msclip_attention_bwd_varlen")
    return dqkv


def embed_tokens_bwd_packed(tokens, dx, cu, demb, dpos):
    """Embedding backward over packed captions: demb += rows (atomics), dpos[l] = fixed-order sum of the captions' rows l."""
    B, L = tokens.shape
    assert dx.dtype == torch.float32 and dx.stride(-1) == 1 and cu.dtype == torch.int32
    _check(lib().msclip_embed_tokens_bwd_packed(_p(tokens), _p(dx), dx.stride(0), _p(cu), _p(demb), _p(dpos), B, L, dx.shape[1],
                                                demb.shape[0], _stream()), "msclip_embed_tokens_bwd_packed")


def l2norm_bwd(x, dy, dx):
    M, E = x.shape
    _check(lib().msclip_l2norm_bwd(_p(x), x.stride(0), _p(dy), dy.stride(0), _p(dx), dx.stride(0), M, E, _stream()),
           "msclip_l2norm_bwd")
    return dx


def clip_loss_bwd_g(S, lse_row, lse_col, label_off, w, G, dscale_part=None):
    R, N = S.shape
    _check(lib().msclip_clip_loss_bwd_g(_p(S), S.stride(0), _p(lse_row), _p(lse_col), label_off, w, _p(G), G.stride(0),
                                        _p(dscale_part), R, N, G.shape[1], _stream()), "msclip_clip_loss_bwd_g")
    return G


def embed_tokens_bwd(tokens, dx, demb, dpos):
    B, L = tokens.shape
    _check(lib().msclip_embed_tokens_bwd(_p(tokens), _p(dx), dx.stride(0), _p(demb), _p(dpos), B, L, dx.shape[1],
                                         demb.shape[0], _stream()), "msclip_embed_tokens_bwd")


def adapter_sum(xin, t, dww, dwb, out, B, L, g, usecls):
    _check(lib().msclip_adapter_sum(_p(xin), xin.stride(0), _p(t), t.stride(0), _p(dww), _p(dwb), _p(out), out.stride(0), B, L,
                                    g, xin.shape[1], int(usecls), _stream()), "msclip_adapter_sum")
    return out


def adapter_dx(dsum, dww, dx, B, L, g, usecls):
    _check(lib().msclip_adapter_dx(_p(dsum), dsum.stride(0), _p(dww), _p(dx), dx.stride(0), B, L, g, dsum.shape[1],
                                   int(usecls), _stream()), "msclip_adapter_dx")
    return dx


def im2col(x, B, H, W, C, KH, KW, stride, pad, image=False, kalign=64):
    """Patch matrix [B*Ho*Wo, Kp] bf16 of an NHWC bf16 activation (or of the NCHW fp32 / bf16 input image), Kp = KH*KW*C
    rounded up to kalign (64: a GEMM's K axis; a multiple of 8 is enough for the token-major weight-gradient GEMM, whose
    operand rows only have to be 16-byte aligned), column (kh*KW + kw)*C + ci."""
    assert kalign % 8 == 0
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    Kp = (KH * KW * C + kalign - 1) // kalign * kalign
    n = B * Ho * Wo * Kp
    flat = torch.empty(n + 128, dtype=torch.bfloat16, device=x.device)     # slack behind the matrix: a 128-channel region read from the last row
    col = flat[:n].view(B * Ho * Wo, Kp)
    kind = (1 if x.dtype == torch.float32 else 2) if image else 0
    if not image:
        _bf16(x)
    _check(lib().msclip_im2col(_p(x), kind, _p(col), B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kp, _stream()), "msclip_im2col")
    return col


IMAGE_WGRAD_BLOCKS = 1280


def image_conv_wgrad_ok(img, dy):
    """Can msclip_image_conv_wgrad take this image / output gradient (fp32 NCHW image of 3 channels, side <= 256 and a multiple
    of 4; <= 64 output channels, a multiple of 8)?"""
    return (img.dtype == torch.float32 and img.dim() == 4 and img.shape[1] == 3 and img.shape[2] == img.shape[3] and
            img.is_contiguous() and img.shape[2] % 4 == 0 and img.shape[2] <= 256 and dy.dtype == torch.bfloat16 and
            dy.stride(1) == 1 and dy.shape[1] % 8 == 0 and dy.shape[1] <= 64 and dy.stride(0) % 8 == 0)


def image_conv_wgrad(img, dy):
    """3 x 3 / stride 2 / pad 1 convolution on the input image: -> (dW fp32 [co, 27] with column (kh * 3 + kw) * 3 + ci, sum of
    dy fp32 [co]) from ONE pass over dy [B * Ho * Ho, co] and the image -- no patch matrix, no separate bias-sum pass."""
    assert image_conv_wgrad_ok(img, dy)
    B, S, co = img.shape[0], img.shape[2], dy.shape[1]
    Ho = (S - 1) // 2 + 1
    assert dy.shape[0] >= B * Ho * Ho
    cp = (co + 15) // 16 * 16
    part = torch.empty(IMAGE_WGRAD_BLOCKS, cp * 32, dtype=torch.float32, device=dy.device)
    _check(lib().msclip_image_conv_wgrad(_p(img), _p(dy), dy.stride(0), _p(part), IMAGE_WGRAD_BLOCKS, B, S, co, _stream()),
           "msclip_image_conv_wgrad")
    both = colsum(part).view(cp, 32)
    return both[:co, :27], both[:co, 27]


def col2im(dcol, dx, B, H, W, C, KH, KW, stride, pad, accumulate=False):
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    _bf16(dcol)
    _bf16(dx)
    _check(lib().msclip_col2im(_p(dcol), dcol.stride(0), _p(dx), B, H, W, C, KH, KW, stride, pad, Ho, Wo, int(accumulate),
                               _stream()), "msclip_col2im")
    return dx


def relu_bwd(dy, y, out=None, dy2=None):
    """(dy [+ dy2]) * (y > 0) on bf16 tensors of equal size."""
    _bf16(dy)
    _bf16(y)
    if out is None:
        out = torch.empty_like(dy)
    assert dy.is_contiguous() and y.is_contiguous() and out.is_contiguous() and dy.numel() == y.numel() == out.numel()
    _check(lib().msclip_relu_bwd(_p(dy), _p(dy2) if dy2 is not None else None, _p(y), _p(out), dy.numel(), _stream()),
           "msclip_relu_bwd")
    return out


def dwpool_bwd(dpool, w, dtop, B, H, W, C, k, accumulate=False):
    _check(lib().msclip_dwpool_bwd(_p(dpool), dpool.stride(0), _p(w), _p(dtop), B, H, W, C, k, int(accumulate), _stream()),
           "msclip_dwpool_bwd")
    return dtop


def dwpool_wgrad(dpool, top, B, H, W, C, k):
    """-> fp32 [k*k, C]: gradient of msclip_dwpool's filter table."""
    S = max(1, min(1024, 4096 // k, B * (H // k) * (W // k)))          # grid (k, S): one block per window row and slab
    part = torch.empty(S, k * k * C, dtype=torch.float32, device=dpool.device)
    _check(lib().msclip_dwpool_wgrad(_p(dpool), dpool.stride(0), _p(top), _p(part), B, H, W, C, k, S, _stream()),
           "msclip_dwpool_wgrad")
    return colsum(part).view(k * k, C)


def dw3x3_wgrad(dsum, x, B, L, g):
    """-> fp32 [9, C]: gradient of the token-grid depthwise 3x3 filter of msclip_adapter_sum."""
    C = dsum.shape[1]
    S = max(1, min(256, B))
    part = torch.empty(S, 9 * C, dtype=torch.float32, device=dsum.device)
    _check(lib().msclip_dw3x3_wgrad(_p(dsum), dsum.stride(0), _p(x), x.stride(0), _p(part), B, L, g, C, S, _stream()),
           "msclip_dw3x3_wgrad")
    return colsum(part).view(9, C)


def _bn_chunks(M):
    return 1 if M <= 2048 else min(512, (M + 1023) // 1024)


def _bn_fold_rows(M, C, *mats):
    """Column reductions over a narrow matrix waste lanes (48 channels = 48 of a wave's 64 lanes, 192-byte rows).  r
    consecutive rows are read as ONE row of r*C columns (contiguous matrices only): every lane works on full cache lines,
    and the r partial results per channel are added afterwards."""
    r = 1
    if all(m.is_contiguous() and m.shape[1] == C for m in mats):
        while C * r < 768 and M % (r * 2) == 0 and M // (r * 2) >= 64:
            r *= 2
    return r


def bn_stats(x, M=None, gamma=None, beta=None, eps=1e-5):
    """x [M, C] bf16 or fp32 -> (mean [C], biased variance [C]) over the rows (fp32); with gamma / beta also
    (rstd, scale = gamma rstd, shift = beta - mean scale): the whole per-channel tail is ONE launch (msclip_bn_finish)."""
    M = x.shape[0] if M is None else M
    C = x.shape[1]
    x = x[:M]
    r = _bn_fold_rows(M, C, x)
    xw = x.view(M // r, C * r) if r > 1 else x
    Mw, Cw = xw.shape
    ch = _bn_chunks(Mw)
    part = torch.empty(ch, 2 * Cw, dtype=torch.float32, device=x.device)
    _check(lib().msclip_bn_stats(_p(xw), xw.stride(0), int(x.dtype == torch.float32), _p(part), Mw, Cw, ch, _stream()),
           "msclip_bn_stats")
    sums = colsum(part) if ch > 1 else part[0]                # [2][r][C]
    plain = gamma is None
    if plain:
        gamma, beta = _bn_unit(C, x.device)
    # every vector tiled r times (out [5][r * C]): bn_apply / bn_bwd over the same r-folded map take the tiled rows as they are
    # (instead of a repeat() launch per vector per pass); the tensors handed back are the first C entries
    out = torch.empty(5, r * C, dtype=torch.float32, device=x.device)
    _check(lib().msclip_bn_finish_tiled(_p(sums), r, C, M, _p(gamma), _p(beta), eps, _p(out), r, _stream()), "msclip_bn_finish_tiled")
    vecs = tuple(out[k, :C] for k in range(5))
    for v in vecs:
        v._bn_tiled = (out, r)
    return vecs[:2] if plain else vecs


_BN_UNIT = {}


def _bn_unit(C, device):
    k = (C, device)
    if k not in _BN_UNIT:
        _BN_UNIT[k] = (torch.ones(C, dtype=torch.float32, device=device), torch.zeros(C, dtype=torch.float32, device=device))
    return _BN_UNIT[k]


def bn_stats_partials(x, chunks=None):
    """part [chunks][2][C] fp32 = (sum x, sum x^2) per row chunk of x [M, C] (msclip_bn_stats without the fold)."""
    M, C = x.shape
    assert x.stride(1) == 1 and x.dtype in (torch.float32, torch.bfloat16)
    ch = chunks if chunks is not None else _bn_chunks(M)
    part = torch.empty(ch, 2 * C, dtype=torch.float32, device=x.device)
    _check(lib().msclip_bn_stats(_p(x), x.stride(0), int(x.dtype == torch.float32), _p(part), M, C, ch, _stream()), "msclip_bn_stats")
    return part


def bn_bwd_token_columns(dy, x, mean, rstd, gamma, dx, L, n_stat):
    """Train-mode BatchNorm backward of a BatchNorm over the rows 1 .. L - 1 of every sample of a token matrix [B L, D], with the
    matrices viewed as [B, L * D] (a sample per row, a (token, channel) pair per column): fp32 dy, x (raw map), dx.  Token 0's
    columns (the class token, which the BatchNorm does not see) run with mean 0, rstd 1, gamma 1 and are left out of the sums, so
    dx = dy there: ONE dx pass writes the whole matrix.  -> (dgamma [D], dbeta [D])."""
    B, LD = dy.shape
    D = LD // L
    for t in (dy, x, dx):
        assert t.dtype == torch.float32 and t.shape == (B, LD) and t.stride(1) == 1
    dev = dy.device
    mean_t, rstd_t = mean.repeat(L), rstd.repeat(L)
    mean_t[:D].zero_()
    rstd_t[:D].fill_(1.0)
    ch = _bn_chunks(B)
    part = torch.empty(ch, 2 * LD, dtype=torch.float32, device=dev)
    st = _stream()
    _check(lib().msclip_bn_bwd_reduce(_p(dy), dy.stride(0), 1, _p(x), x.stride(0), 1, _p(mean_t), _p(rstd_t), _p(part), B, LD, ch, st),
           "msclip_bn_bwd_reduce")
    part.view(ch, 2, L, D)[:, :, 0].zero_()                 # the class token's columns take no part in the sums
    tail = torch.empty(3, LD, dtype=torch.float32, device=dev)          # (dbeta, dgamma, gamma), each tiled L times
    _check(lib().msclip_bn_bwd_finish(_p(part), ch, L, D, _p(gamma), _p(tail), st), "msclip_bn_bwd_finish")
    dbeta, dgamma = tail[0, D:2 * D].clone(), tail[1, D:2 * D].clone()
    tail.view(3, L, D)[:2, 0].zero_()
    tail.view(3, L, D)[2, 0].fill_(1.0)
    _check(lib().msclip_bn_bwd_dx(_p(dy), dy.stride(0), 1, _p(x), x.stride(0), 1, _p(mean_t), _p(rstd_t), _p(tail[2]), _p(tail[0]),
                                  _p(tail[1]), _p(dx), dx.stride(0), B, LD, n_stat, st), "msclip_bn_bwd_dx")
    return dgamma, dbeta


def bn_finish(sums, C, n, gamma, beta, eps, out):
    """sums [2][C] = (sum x, sum x^2) over n rows -> out [5][C] = mean, biased variance, rstd, scale = gamma rstd, shift = beta - mean
    scale (msclip_bn_finish_tiled, no row fold)."""
    _f32(sums); _f32(out)
    assert sums.numel() >= 2 * C and tuple(out.shape) == (5, C) and out.is_contiguous()
    _check(lib().msclip_bn_finish_tiled(_p(sums), 1, C, n, _p(gamma), _p(beta), eps, _p(out), 1, _stream()), "msclip_bn_finish_tiled")
    return out


def bn_apply(x, scale, shift, out, M=None, relu=False, resid=None):
    M = x.shape[0] if M is None else M
    C = x.shape[1]
    xs, os_, rs = x[:M], out[:M], (resid[:M] if resid is not None else None)
    r = _bn_fold_rows(M, C, *([xs, os_] + ([rs] if rs is not None else [])))
    if r > 1:                                            # narrow maps: r rows as one row of r*C columns (full cache lines per wave)
        xs, os_ = xs.view(M // r, C * r), os_.view(M // r, C * r)
        rs = rs.view(M // r, C * r) if rs is not None else None
        t = getattr(scale, "_bn_tiled", None)
        if t is not None and t[1] == r and getattr(shift, "_bn_tiled", (None,))[0] is t[0]:
            scale, shift = t[0][3], t[0][4]              # bn_stats' own tiled rows
        else:
            scale, shift = scale.repeat(r), shift.repeat(r)
    _check(lib().msclip_bn_apply(_p(xs), xs.stride(0), int(x.dtype == torch.float32), _p(scale), _p(shift),
                                 _p(rs) if rs is not None else None, rs.stride(0) if rs is not None else 0,
                                 _p(os_), os_.stride(0), int(out.dtype == torch.float32), M // r, C * r, int(relu), _stream()),
           "msclip_bn_apply")
    return out


def bn_bwd(dy, x, mean, rstd, gamma, dx, M=None):
    """Train-mode BatchNorm backward on matrices [M, C] (x: bf16 or fp32; dy and dx: one dtype, bf16 or fp32):
    -> (dgamma, dbeta); dx is filled."""
    M = x.shape[0] if M is None else M
    C = x.shape[1]
    assert dy.dtype == dx.dtype
    xf, df = int(x.dtype == torch.float32), int(dy.dtype == torch.float32)
    x, dy = x[:M], dy[:M]
    r = _bn_fold_rows(M, C, x, dy)
    xw, dyw = (x.view(M // r, C * r), dy.view(M // r, C * r)) if r > 1 else (x, dy)
    t = getattr(mean, "_bn_tiled", None)
    if r > 1 and t is not None and t[1] == r and getattr(rstd, "_bn_tiled", (None,))[0] is t[0]:
        mw, rw = t[0][0], t[0][2]                        # bn_stats' own tiled rows
    else:
        mw, rw = (mean.repeat(r), rstd.repeat(r)) if r > 1 else (mean, rstd)
    Mw, Cw = xw.shape
    ch = _bn_chunks(Mw)
    part = torch.empty(ch, 2 * Cw, dtype=torch.float32, device=x.device)
    _check(lib().msclip_bn_bwd_reduce(_p(dyw), dyw.stride(0), df, _p(xw), xw.stride(0), xf, _p(mw), _p(rw), _p(part), Mw, Cw,
                                      ch, _stream()), "msclip_bn_bwd_reduce")
    # the per-channel tail in ONE launch: chunk / fold sums, and (dbeta, dgamma, gamma) tiled r times for the dx pass
    tail = torch.empty(3, r * C, dtype=torch.float32, device=x.device)
    _check(lib().msclip_bn_bwd_finish(_p(part), ch, r, C, _p(gamma), _p(tail), _stream()), "msclip_bn_bwd_finish")
    dbeta, dgamma = tail[0, :C], tail[1, :C]
    dxs = dx[:M]
    if r > 1 and dxs.is_contiguous():
        dxw = dxs.view(M // r, C * r)
        _check(lib().msclip_bn_bwd_dx(_p(dyw), dyw.stride(0), df, _p(xw), xw.stride(0), xf, _p(mw), _p(rw), _p(tail[2]), _p(tail[0]),
                                      _p(tail[1]), _p(dxw), dxw.stride(0), Mw, Cw, M, _stream()), "msclip_bn_bwd_dx")
    else:
        _check(lib().msclip_bn_bwd_dx(_p(dy), dy.stride(0), df, _p(x), x.stride(0), xf, _p(mean), _p(rstd), _p(gamma),
                                      _p(dbeta), _p(dgamma), _p(dx), dx.stride(0), M, C, M, _stream()), "msclip_bn_bwd_dx")
    return dgamma, dbeta


def bn_bwd_fused_ok(dy, sides, y=None, dy2=None, M=None):
    """Shapes msclip_bn_bwd_fused takes: bf16 contiguous [M, C] gradients / mask map, contiguous raw maps (fp32; or bf16 = the
    normalised values of the two-pass forward, all sides alike), C % 4 == 0."""
    M = sides[0][0].shape[0] if M is None else M
    C = sides[0][0].shape[1]
    mats = [dy] + [t for t in (y, dy2) if t is not None]
    if C % 4 or len(sides) not in (1, 2) or any(t.dtype != torch.bfloat16 or t.shape[1] != C or not t.is_contiguous() or t.shape[0] < M
                                                  for t in mats):
        return False
    return all(x.dtype == sides[0][0].dtype and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() and x.shape[1] == C and
               x.shape[0] >= M and dx.dtype == torch.bfloat16 and dx.is_contiguous() and dx.shape[1] == C and dx.shape[0] >= M
               for x, _, _, _, dx in sides)


def bn_bwd_fused(dy, sides, y=None, dy2=None, M=None, chunks=None, dx_chunks=0):
    """Train-mode BatchNorm backward of one or two BatchNorms behind the same upstream gradient d = bf16(dy [+ dy2]) * (y > 0)
    (msclip_bn_bwd_fused: no msclip_relu_bwd pass, no map of d; four columns per thread).  sides = [(x_raw fp32 [M, C], mean, rstd,
    gamma, dx bf16 [M, C]), ...] -> [(dgamma, dbeta), ...]; every dx is filled."""
    M = sides[0][0].shape[0] if M is None else M
    C = sides[0][0].shape[1]
    assert bn_bwd_fused_ok(dy, sides, y, dy2, M)
    mats = [dy[:M]] + [t[:M] for t in (y, dy2) if t is not None] + [x[:M] for x, *_ in sides] + [s[4][:M] for s in sides]
    r = _bn_fold_rows(M, C, *mats)
    Mw, Cw = M // r, C * r
    # row chunks of the reduce pass: 128-512 rows each, at least ~200 of them where the map has the rows (sweep of
    # tools/probes/bn_bwd_fused_bench.py: 14^2 / 7^2 maps lose 2 x with the 1 024-row chunks of the scalar kernels; beyond 784 chunks
    # msclip_bn_bwd_finish -- a workgroup per channel walks every chunk's partials -- costs more than the reduce pass gains)
    ch = max(1, min(784, max(Mw // 256, 196), Mw // 32))
    if chunks:
        ch = chunks                                      # (probe: tools/probes/bn_bwd_fused_bench.py)
    dev = dy.device
    sd, keep = [], []
    for x, mean, rstd, gamma, dx in sides:
        t = getattr(mean, "_bn_tiled", None)
        if r > 1 and t is not None and t[1] == r and getattr(rstd, "_bn_tiled", (None,))[0] is t[0]:
            mw, rw = t[0][0], t[0][2]                    # bn_stats' own tiled rows
        else:
            mw, rw = (mean.repeat(r), rstd.repeat(r)) if r > 1 else (mean.contiguous(), rstd.contiguous())
        part = torch.empty(ch, 2 * Cw, dtype=torch.float32, device=dev)
        tail = torch.empty(3, Cw, dtype=torch.float32, device=dev)
        keep.append((mw, rw, part, tail))
        sd.append(BnBwdSide(_p(x), Cw, _p(mw), _p(rw), _p(tail[2]), _p(tail[0]), _p(tail[1]), _p(dx), Cw, _p(part),
                            int(x.dtype == torch.bfloat16)))
    s2 = ctypes.byref(sd[1]) if len(sd) == 2 else None
    args = (_p(dy), Cw, _p(dy2) if dy2 is not None else None, Cw, _p(y) if y is not None else None, Cw, ctypes.byref(sd[0]), s2, Mw, Cw)
    st = _stream()
    _check(lib().msclip_bn_bwd_fused(0, *args, ch, M, st), "msclip_bn_bwd_fused (reduce)")
    out = []
    for (x, mean, rstd, gamma, dx), (mw, rw, part, tail) in zip(sides, keep):
        _check(lib().msclip_bn_bwd_finish(_p(part), ch, r, C, _p(gamma), _p(tail), st), "msclip_bn_bwd_finish")
        out.append((tail[1, :C], tail[0, :C]))
    _check(lib().msclip_bn_bwd_fused(1, *args, dx_chunks, M, st), "msclip_bn_bwd_fused (dx)")
    return out


def bn_fold_bwd(G, w_raw, dshift, gamma, mean, var, eps):
    """Frozen-statistics BatchNorm fold, backward: G = dL/d(folded filter) [cout, ...] (fp32, rows contiguous), w_raw the raw
    filter of the same shape -> (dW like w_raw, dgamma [cout], dbeta [cout])."""
    cout = w_raw.shape[0]
    K = w_raw.numel() // cout
    Gm = G.reshape(cout, -1)
    if Gm.shape[1] > 1 and Gm.stride(1) != 1:
        Gm = Gm.contiguous()
    assert Gm.dtype == torch.float32 and Gm.shape[1] == K and w_raw.dtype == torch.float32 and w_raw.is_contiguous()
    dW = torch.empty_like(w_raw)
    dgb = torch.empty(2, cout, dtype=torch.float32, device=w_raw.device)
    _check(lib().msclip_bn_fold_bwd(_p(Gm), Gm.stride(0), _p(w_raw), cout, K, _p(dshift), _p(gamma), _p(mean), _p(var), eps, _p(dW),
                                    _p(dgb[0]), _p(dgb[1]), _stream()), "msclip_bn_fold_bwd")
    return dW, dgb[0], dgb[1]


class PackItem(ctypes.Structure):
    """msclip_pack_item (include/msclip_hip.h)."""
    _fields_ = [("w", ctypes.c_void_p), ("g", ctypes.c_void_p), ("b", ctypes.c_void_p), ("mu", ctypes.c_void_p), ("var", ctypes.c_void_p),
                ("w2", ctypes.c_void_p), ("g2", ctypes.c_void_p), ("b2", ctypes.c_void_p), ("mu2", ctypes.c_void_p), ("var2", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("bias_out", ctypes.c_void_p), ("eps", ctypes.c_float), ("eps2", ctypes.c_float),
                ("co", ctypes.c_int), ("ci", ctypes.c_int), ("kh", ctypes.c_int), ("kw", ctypes.c_int), ("kpad", ctypes.c_int),
                ("mode", ctypes.c_int), ("col0", ctypes.c_int), ("ld", ctypes.c_int), ("bias_mode", ctypes.c_int), ("bias_col0", ctypes.c_int)]


class PackPlan:
    """Device-resident item table of msclip_pack_weights: every derived conv-side tensor described once (sources = the module's
    parameter storage, destinations = the engine's persistent operand tensors), re-run after every optimizer step."""

    def __init__(self, device):
        self.device, self.items, self.keep = device, [], []

    def add(self, w, out=None, *, bn=None, eps=1e-5, w2=None, bn2=None, eps2=1e-5, mode=0, col0=0, ld=0, bias_out=None, bias_mode=0,
            bias_col0=0, co=None):
        """w fp32 [co, ci, kh, kw] (contiguous parameter storage); bn / bn2 = (gamma, beta, running_mean, running_var)."""
        it = PackItem()
        w4 = w if w.dim() == 4 else w.view(w.shape[0], -1, 1, 1)
        assert w4.dtype == torch.float32 and w4.is_contiguous()
        it.w = w4.data_ptr()
        it.co, it.ci, it.kh, it.kw = (co if co is not None else w4.shape[0]), w4.shape[1], w4.shape[2], w4.shape[3]
        tensors = [w4]
        if bn is not None:
            for t in bn:
                assert t.dtype == torch.float32 and t.is_contiguous()
            it.g, it.b, it.mu, it.var = (t.data_ptr() for t in bn)
            it.eps = eps
            tensors += list(bn)
        if w2 is not None:
            assert w2.dtype == torch.float32 and w2.is_contiguous() and w2.numel() == it.co * it.ci
            it.w2 = w2.data_ptr()
            tensors.append(w2)
        if bn2 is not None:
            it.g2, it.b2, it.mu2, it.var2 = (t.data_ptr() for t in bn2)
            it.eps2 = eps2
            tensors += list(bn2)
        if out is not None:
            assert out.is_contiguous() or mode == 1
            it.out = out.data_ptr()
            it.mode, it.col0, it.ld = mode, col0, ld
            if mode == 0:
                assert out.dtype == torch.bfloat16 and out.shape[0] == it.co and out.shape[1] >= it.ci * it.kh * it.kw
                it.kpad = out.shape[1]
            else:
                assert out.dtype == torch.float32 and ld >= col0 + it.co
            tensors.append(out)
        if bias_out is not None:
            assert bias_out.dtype == torch.float32 and bias_out.is_contiguous() and bias_mode in (1, 2, 3)
            assert bias_mode != 3 or (out is None and it.ci <= 1024)
            it.bias_out, it.bias_mode, it.bias_col0 = bias_out.data_ptr(), bias_mode, bias_col0
            tensors.append(bias_out)
        self.items.append(it)
        self.keep.append(tensors)

    def finalize(self):
        n = len(self.items)
        arr = (PackItem * n)(*self.items)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        self.table = raw.to(self.device)
        starts, tot = [], 0
        for it in self.items:
            starts.append(tot)
            rowlen = it.kpad if it.mode == 0 else it.ci * it.kh * it.kw
            if it.out:
                tot += max(1, it.co * -(-rowlen // 1024))
            else:
                tot += -(-it.co // 8) if it.bias_mode == 3 else 1         # the matrix-vector bias: 8 outputs per workgroup
        starts.append(tot)
        self.blk_start = torch.tensor(starts, dtype=torch.int32).to(self.device)
        self.n_items, self.n_blocks = n, tot
        return self

    def run(self):
        _check(lib().msclip_pack_weights(_p(self.table), _p(self.blk_start), self.n_items, self.n_blocks, _stream()), "msclip_pack_weights")


class TransposeItem(ctypes.Structure):
    """msclip_transpose_item (include/msclip_hip.h)."""
    _fields_ = [("in_", ctypes.c_void_p), ("out", ctypes.c_void_p), ("ldi", ctypes.c_int), ("ldo", ctypes.c_int), ("M", ctypes.c_int),
                ("C", ctypes.c_int)]


class TransposePlan:
    """W^T of many bf16 matrices in ONE launch (msclip_transpose_bf16_multi): the sources' storage must persist (the engine's
    packed block weights, rewritten in place by the optimizer kernel); the outputs [C, M] are owned by the plan and rewritten
    by every run() -- on the stream run() is called on, so whoever still reads the previous run's outputs must be ordered in
    front of it.  `key` = the sources' addresses (a plan is valid while they do not move)."""

    def __init__(self, tensors):
        self.device = tensors[0].device
        self.key = tuple(t.data_ptr() for t in tensors)
        items, starts, tot, self.outs = [], [], 0, []
        for t in tensors:
            _bf16(t)
            M, C = t.shape
            assert M % 64 == 0 and C % 8 == 0 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0
            out = torch.empty(C, M, dtype=torch.bfloat16, device=self.device)
            it = TransposeItem()
            it.in_, it.out, it.ldi, it.ldo, it.M, it.C = t.data_ptr(), out.data_ptr(), t.stride(0), M, M, C
            items.append(it)
            self.outs.append(out)
            starts.append(tot)
            tot += (M // 64) * ((C + 63) // 64)
        starts.append(tot)
        arr = (TransposeItem * len(items))(*items)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(self.device)
        self.blk_start = torch.tensor(starts, dtype=torch.int32).to(self.device)
        self.n_items, self.n_blocks = len(items), tot

    def run(self):
        _check(lib().msclip_transpose_bf16_multi(_p(self.table), _p(self.blk_start), self.n_items, self.n_blocks, _stream()),
               "msclip_transpose_bf16_multi")
        return self.outs


class AdamwTensor(ctypes.Structure):
    """msclip_adamw_tensor (include/msclip_hip.h)."""
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_longlong), ("lr", ctypes.c_float), ("weight_decay", ctypes.c_float),
                ("pk", ctypes.c_void_p), ("pk_scale", ctypes.c_float), ("pk_f32", ctypes.c_int)]


class AdamwPlan:
    """The host-side tensor table of msclip_adamw_multi, built once and reused while the tensors stay where they are (the
    training step's parameters, its gradient slots inside the all-reduce buckets and the optimizer state do).
    items: [(p, g, m, v, lr, weight_decay)] or [(p, g, m, v, lr, weight_decay, packed, packed_scale)] of contiguous fp32
    tensors (flat views are fine); `packed` = a bf16 or fp32 tensor of the same element count that receives
    p_new * packed_scale from the same kernel (the engine's operand copy), or None."""

    def __init__(self, items):
        n = len(items)
        self.n = n
        self.arr = (AdamwTensor * max(n, 1))()
        self.device = items[0][0].device if n else None
        # parameters, moments and packed copies live as long as the table; the gradients do not (set_grads re-points them)
        self.keep = [(it[0], it[2], it[3]) + tuple(it[6:7]) for it in items]
        for a, it in zip(self.arr, items):
            p, g, m, v, lr, wd = it[:6]
            for t in (p, g, m, v):
                assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel() and t.device == p.device
            a.p, a.g, a.m, a.v, a.n, a.lr, a.weight_decay = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, wd
            pk = it[6] if len(it) > 6 else None
            if pk is not None:
                assert pk.dtype in (torch.bfloat16, torch.float32) and pk.is_contiguous() and pk.numel() == p.numel() and pk.device == p.device
                a.pk, a.pk_scale, a.pk_f32 = pk.data_ptr(), float(it[7]), int(pk.dtype == torch.float32)

    def set_grads(self, ptrs):
        """Device addresses of this step's gradients, one per item (same element counts as at construction)."""
        for a, g in zip(self.arr, ptrs):
            a.g = g

    def set_rates(self, rates):
        """rates: [(lr, weight_decay)] per item."""
        for a, (lr, wd) in zip(self.arr, rates):
            a.lr, a.weight_decay = lr, wd

    def run(self, beta1, beta2, eps, step):
        if not self.n:
            return
        with torch.cuda.device(self.device):
            _check(lib().msclip_adamw_multi(self.arr, self.n, beta1, beta2, eps, step, _stream()), "msclip_adamw_multi")


def adamw_multi(items, beta1, beta2, eps, step):
    """items: see AdamwPlan: one msclip_adamw_multi call (a handful of launches for the model's 325 tensors instead of one
    each)."""
    AdamwPlan(items).run(beta1, beta2, eps, step)


def adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    _check(lib().msclip_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step, _stream()),
           "msclip_adamw")
