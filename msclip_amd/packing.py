"""Weight packing for the HIP path: fold eval-mode BatchNorm, merge the stem's
1x1 shortcut into the 3x3 centre tap, lay weights out as [N][Kpad] bf16 for the
MFMA GEMM, and build the implicit-GEMM K-chunk tables.

Pure tensor algebra (device-agnostic) so tests can check every fold on CPU
against the oracle.  Reference lines: BN eps 1e-5 everywhere except
ConvResBlock's 1e-6 (lib/models/clip_openai_pe_res_v1.py:1825-1840); stem stage
= relu(BN(conv3x3) + BN(conv1x1)) (ibid. 1920-1936).
"""
import torch

BK = 64


def round_up(x, m):
    return (x + m - 1) // m * m


_FOLDS = "__bn_folds__"


def bn_fold_all(sd):
    """Every BatchNorm of the state dict folded in five multi-tensor launches per eps class instead of five small kernels
    per layer (the engine re-packs after every optimizer step: 36 layers x 5 dependent launches were ~2 ms on the GPU
    timeline).  Stored in sd under a private key; bn_fold() looks there first.  eps: 1e-6 inside the ConvResBlocks
    (M.py:1825-1861), 1e-5 elsewhere."""
    groups = {}
    for k in sd:
        if k.endswith(".running_var") and sd[k].is_cuda:
            p = k[:-len(".running_var")]
            eps = 1e-6 if ".resnet_stage.conv_" in p and "parallel_branch" in p else 1e-5
            groups.setdefault(eps, []).append(p)
    folds = {}
    for eps, ps in groups.items():
        w = [sd[p + ".weight"].float() for p in ps]
        b = [sd[p + ".bias"].float() for p in ps]
        mu = [sd[p + ".running_mean"].float() for p in ps]
        den = torch._foreach_add([sd[p + ".running_var"].float() for p in ps], eps)
        torch._foreach_sqrt_(den)
        scale = torch._foreach_div(w, den)
        shift = torch._foreach_sub(b, torch._foreach_mul(mu, scale))
        for p, sc, sh in zip(ps, scale, shift):
            folds[(p, eps)] = (sc, sh)
    sd[_FOLDS] = folds


def bn_fold(sd, prefix, eps):
    """-> (scale, shift) with BN(x) = scale * x + shift in eval mode."""
    pre = sd.get(_FOLDS)
    if pre is not None and (prefix, eps) in pre:
        return pre[(prefix, eps)]
    w, b = sd[prefix + ".weight"].float(), sd[prefix + ".bias"].float()
    mu, var = sd[prefix + ".running_mean"].float(), sd[prefix + ".running_var"].float()
    scale = w / torch.sqrt(var + eps)
    return scale, b - mu * scale


def conv_weight_matrix(w):
    """[Cout, Cin, KH, KW] -> [Cout, KH*KW*Cin] with K index (kh*KW + kw)*Cin + ci (NHWC gather order)."""
    co, ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci)


def pad_k(wm):
    n, k = wm.shape
    kp = round_up(k, BK)
    if kp == k:
        return wm.contiguous()
    out = wm.new_zeros(n, kp)
    out[:, :k] = wm
    return out


def ktab(kh, kw, cin, w_in):
    """K-chunk table of the implicit-GEMM loader (include/msclip_hip.h, mode 1)."""
    assert cin % 8 == 0
    cpt = cin // 8
    k = kh * kw * cin
    kp = round_up(k, BK)
    tab = torch.full((kp // 8,), -1, dtype=torch.int32)
    for c in range(k // 8):
        tap, ch = divmod(c, cpt)
        y, x = divmod(tap, kw)
        doff = (y * w_in + x) * cin + ch * 8
        assert doff < (1 << 20) and y < 16 and x < 16
        tab[c] = doff | (y << 20) | (x << 24)
    return tab


_KTAB_ON_DEVICE = {}


def ktab_on(device, kh, kw, cin, w_in):
    """The chunk table of a geometry, built and uploaded once per device: re-packing after an optimizer step must not
    issue host-to-device copies (a pageable upload blocks the host until the stream has drained)."""
    key = (str(device), kh, kw, cin, w_in)
    t = _KTAB_ON_DEVICE.get(key)
    if t is None:
        t = _KTAB_ON_DEVICE[key] = ktab(kh, kw, cin, w_in).to(device)
    return t


class ConvSpec:
    """One convolution lowered to the gathering GEMM: packed weight, bias, chunk table, geometry."""

    def __init__(self, w_oihw, bias, h_in, w_in, stride, pad):
        co, ci, kh, kw = w_oihw.shape
        self.cout, self.cin, self.kh, self.kw = co, ci, kh, kw
        self.h_in, self.w_in, self.stride, self.pad = h_in, w_in, stride, pad
        self.h_out = (h_in + 2 * pad - kh) // stride + 1
        self.w_out = (w_in + 2 * pad - kw) // stride + 1
        self.weight = pad_k(conv_weight_matrix(w_oihw.float())).to(torch.bfloat16)
        self.bias = bias.float().contiguous()
        self.ktab = ktab_on(self.weight.device, kh, kw, ci, w_in)

    def to(self, device):
        self.weight, self.bias = self.weight.to(device), self.bias.to(device)
        self.ktab = ktab_on(device, self.kh, self.kw, self.cin, self.w_in)
        return self

    def geometry(self):
        return (self.h_in, self.w_in, self.cin, self.h_out, self.w_out, self.stride, self.pad)


def stem_dual_weights(sd, stem_prefix, par_prefix):
    """First convs of the stem and of the parallel branch as one [27][2*C1] fp32 filter bank + bias."""
    ws, wp = sd[stem_prefix + ".conv1.weight"].float(), sd[par_prefix + ".conv.weight"].float()
    ss, bs = bn_fold(sd, stem_prefix + ".bn1", 1e-5)
    sp, bp = bn_fold(sd, par_prefix + ".bn", 1e-5)
    w = torch.cat([ws * ss[:, None, None, None], wp * sp[:, None, None, None]], 0)   # [2*C1, 3, 3, 3]
    w = w.reshape(w.shape[0], 27).t().contiguous()                                    # [(ci,kh,kw), 2*C1]
    return w, torch.cat([bs, bp]).contiguous()


def stem_stage(sd, prefix, h_in, stride):
    """relu(BN(conv3x3 s) + BN(conv1x1 s)) == relu(conv3x3' s + bias'): the 1x1/s/p0 shortcut samples exactly the
    centre tap of the 3x3/s/p1 window."""
    w3 = sd[prefix + ".conv1.weight"].float()
    w1 = sd[prefix + ".downsample.0.weight"].float()
    s3, b3 = bn_fold(sd, prefix + ".bn1", 1e-5)
    s1, b1 = bn_fold(sd, prefix + ".downsample.1", 1e-5)
    w = w3 * s3[:, None, None, None]
    w[:, :, 1, 1] += w1[:, :, 0, 0] * s1[:, None]
    return ConvSpec(w, b3 + b1, h_in, h_in, stride, 1)


def bottleneck(sd, prefix, h_in, stride, kernel=3, pad=1):
    """ConvResBlock with projection shortcut -> four ConvSpecs (conv1, conv2, residual, conv3), BN eps 1e-6."""
    def folded(conv, bn):
        w = sd[f"{prefix}.{conv}.weight"].float()
        s, b = bn_fold(sd, f"{prefix}.{bn}", 1e-6)
        return w * s[:, None, None, None], b
    w1, b1 = folded("conv1", "bn1")
    w2, b2 = folded("conv2", "bn2")
    w3, b3 = folded("conv3", "bn3")
    wr, br = folded("residual_conv", "residual_bn")
    c1 = ConvSpec(w1, b1, h_in, h_in, 1, 0)
    c2 = ConvSpec(w2, b2, h_in, h_in, stride, pad)
    cr = ConvSpec(wr, br, h_in, h_in, stride, 0)
    c3 = ConvSpec(w3, b3, c2.h_out, c2.w_out, 1, 0)
    return c1, c2, cr, c3


def adapter_weights(sd, prefix, grid):
    """Lateral adapter: depthwise patch-pool filter [k*k][C] (BN scale folded), pointwise ConvSpec whose bias
    carries the BN shift (W_pw @ shift), depthwise 3x3 on the token grid [9][C] + shift."""
    wd = sd[prefix + ".top2bottom_dw_conv.conv.weight"].float()          # [C, 1, k, k]
    sc, sh = bn_fold(sd, prefix + ".top2bottom_dw_conv.bn", 1e-5)
    c, _, k, _ = wd.shape
    pool = (wd[:, 0] * sc[:, None, None]).reshape(c, k * k).t().contiguous()   # [k*k, C]
    wp = sd[prefix + ".top2bottom_pw_conv.conv.weight"].float()           # [D, C, 1, 1]
    pw = ConvSpec(wp, wp[:, :, 0, 0] @ sh, grid, grid, 1, 0)
    wb = sd[prefix + ".bottom_dw_conv.conv.weight"].float()               # [D, 1, 3, 3]
    sb, hb = bn_fold(sd, prefix + ".bottom_dw_conv.bn", 1e-5)
    dww = (wb[:, 0] * sb[:, None, None]).reshape(wb.shape[0], 9).t().contiguous()  # [9, D]
    return pool, k, pw, dww, hb.contiguous()


def qkv_weights(in_w, in_b, heads):
    """in_proj with q pre-scaled by head_dim^-0.5 (a power of two for head_dim 64: exact in bf16)."""
    d = in_w.shape[1]
    scale = float(d // heads) ** -0.5
    w, b = in_w.float().clone(), in_b.float().clone()
    w[:d] *= scale
    b[:d] *= scale
    return w.to(torch.bfloat16).contiguous(), b.contiguous()
