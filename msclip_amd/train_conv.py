"""Backward pass of the convolutional side of the vision tower (second slice): the EarlyconvRes
stem (M.py:1898-2000), the parallel convolutional branch (M.py:1812-1895) and the convolutional halves of the lateral
adapters (M.py:1752-1778).

BatchNorm semantics: FROZEN statistics.  The HIP forward folds every BatchNorm's running mean / variance into the
adjacent convolution (packing.py), and this backward differentiates exactly that function: gamma, beta and the
convolution weights receive gradients, the running statistics are constants (what autograd of the reference gives in
eval() mode -- the mode tests/golden/*.grads.npz was captured in).  Train-mode batch statistics are not implemented.

Every convolution is a GEMM in the forward (implicit im2col gather); here
    dWf = dY^T . im2col(X)      msclip_im2col + the split-K wgrad GEMM of train._wgrad
    dX  = col2im(dY . Wf)       msclip_gemm + msclip_col2im (pointwise convs: the GEMM writes dX directly)
    dbf = column sums of dY
on the FOLDED weight Wf = W * s, bf = beta - mean * s, s = gamma / sqrt(var + eps); the chain rule back to the module's
own parameters is a few per-channel vector operations (fold_grads).  The stem's 1x1 shortcut lives in the centre tap of
its 3x3 (packing.stem_stage): its gradient is the centre tap of the merged filter's gradient.

Activations are read from the engine's workspace (the training forward runs the conv side layer by layer with
engine.force_unfused, so every intermediate map is there): valid until the next forward on the same batch shape.
"""
import torch

from . import hip

BF = torch.bfloat16
F32 = torch.float32


def _zbuf(rows, cols, dev):
    """bf16 [rows, cols] with 64 zero elements of slack behind it (GEMM operands whose K is padded to 64 read past a
    row's end against zero weights: the bytes must be finite)."""
    n = rows * cols
    return torch.zeros(n + 64, dtype=BF, device=dev)[:n].view(rows, cols)


class _Fold:
    """One BatchNorm's frozen statistics: s = gamma * rstd, shift = beta - mean * s."""

    def __init__(self, sd, prefix, eps):
        self.prefix = prefix
        g, b = sd[prefix + ".weight"].float(), sd[prefix + ".bias"].float()
        self.mean = sd[prefix + ".running_mean"].float()
        self.rstd = torch.rsqrt(sd[prefix + ".running_var"].float() + eps)
        self.s = g * self.rstd
        self.shift = b - self.mean * self.s

    def grads(self, out, conv_key, G, w_raw, dshift):
        """G = dL/d(folded filter) in the raw filter's shape, dshift = dL/d(shift) -> raw conv / gamma / beta gradients."""
        sh = (-1,) + (1,) * (G.dim() - 1)
        out[conv_key] = G * self.s.view(sh)
        ds = (G * w_raw).flatten(1).sum(1) - self.mean * dshift
        out[self.prefix + ".weight"] = ds * self.rstd
        out[self.prefix + ".bias"] = dshift.clone()


class ConvSideBackward:
    def __init__(self, train_step):
        self.ts = train_step
        self.e = train_step.eng
        self._wt = {}

    # ------------------------------------------------------------------ generic pieces
    def _w_t(self, key, weight, cout, cin_cols):
        """Folded forward weight [cout, Kp] bf16 -> [Kp or cin_cols, cout padded to 64] bf16: the dgrad GEMM's W operand."""
        t = self._wt.get(key)
        if t is None:
            cp = (cout + 63) // 64 * 64
            t = torch.zeros(cin_cols, cp, dtype=BF, device=weight.device)
            t[:, :cout] = weight[:, :cin_cols].t()
            self._wt[key] = t
        return t

    def _conv_bwd(self, key, spec, x_in, dpre, B, need_dx=True, col=None):
        """dpre: bf16 [B*Ho*Wo, cout] (with slack) -> (G [cout, cin, kh, kw] fp32 wrt the folded filter, dbias [cout],
        dx NHWC bf16 [B*H*W, cin] or None)."""
        from .train import _wgrad
        co, ci, kh, kw = spec.cout, spec.cin, spec.kh, spec.kw
        pix = B * spec.h_out * spec.w_out
        K = kh * kw * ci
        pointwise = (kh, kw, spec.stride, spec.pad) == (1, 1, 1, 0)
        if col is None:
            col = x_in[:pix] if pointwise else hip.im2col(x_in, B, spec.h_in, spec.w_in, ci, kh, kw, spec.stride, spec.pad)
        dwf = _wgrad(dpre, col, pix)[:, :K]
        G = dwf.reshape(co, kh, kw, ci).permute(0, 3, 1, 2).contiguous()
        db = hip.colsum(dpre, M=pix)
        dx = None
        if need_dx:
            if pointwise:
                wt = self._w_t(key, spec.weight, co, ci)
                dx = _zbuf(pix, ci, dpre.device)
                hip.gemm(dpre, wt, dx, M=pix, N=ci, ldx=co)
            else:
                kp = spec.weight.shape[1]
                wt = self._w_t(key, spec.weight, co, kp)
                dcol = torch.empty(pix, kp, dtype=BF, device=dpre.device)
                hip.gemm(dpre, wt, dcol, M=pix, N=kp, ldx=co)
                dx = _zbuf(B * spec.h_in * spec.w_in, ci, dpre.device)
                hip.col2im(dcol, dx, B, spec.h_in, spec.w_in, ci, kh, kw, spec.stride, spec.pad)
                del dcol
        return G, db, dx

    def _relu_bwd(self, dy, y, dy2=None):
        out = _zbuf(dy.shape[0], dy.shape[1], dy.device)
        hip.relu_bwd(dy, y[:dy.shape[0]], out, dy2=dy2)
        return out

    def begin(self, img, w, Bi):
        """Called once per backward: the model's raw parameters and the image patch matrix both Cin = 3 convs share."""
        self.sd = {k: t.detach() for k, t in self.e.model.state_dict().items()}
        self.w, self.Bi, self.img = w, Bi, img
        self.col_img = None
        self.dpar = [None, None]                     # gradient of par[j] handed down to stage j: [shortcut path, conv1 path]

    def _image_cols(self):
        if self.col_img is None:
            S = self.e.S
            self.col_img = hip.im2col(self.img, self.Bi, S, S, 3, 3, 3, 2, 1, image=True)
        return self.col_img

    def _first_conv(self, grads, conv_key, bn_prefix, dpre):
        """wgrad of a 3x3 / stride 2 convolution on the input image (no input gradient)."""
        from .train import _wgrad
        pix = self.Bi * self.e.h1 * self.e.h1
        co = dpre.shape[1]
        dwf = _wgrad(dpre, self._image_cols(), pix)[:, :27]
        G = dwf.reshape(co, 3, 3, 3).permute(0, 3, 1, 2).contiguous()          # (kh, kw, ci) -> [co, ci, kh, kw]
        _Fold(self.sd, bn_prefix, 1e-5).grads(grads, conv_key, G, self.sd[conv_key].float(), hip.colsum(dpre, M=pix))

    # ------------------------------------------------------------------ lateral adapter j + parallel stage j
    def adapter(self, grads, j, dsum, x_pre):
        """dsum: fp32 [Mv, D] gradient of the adapter's pre-LayerNorm sum; x_pre: the token matrix the adapter read."""
        e, w, Bi, sd = self.e, self.w, self.Bi, self.sd
        a = e.adapters[j]
        g, D, C, k, hw = e.g, e.D, a["C"], a["k"], e.par_hw[j]
        g2 = g * g
        p = f"visual.transformer.parallel_lateral_adapter.{j}"
        dT = dsum.view(Bi, e.Lv, D)[:, 1:].reshape(Bi * g2, D)                   # grid rows (the cls row has no top-down term)
        cs = hip.colsum(dT)
        dT_bf = hip.cast_bf16(dT)
        # bottom depthwise 3x3 + BN on the token grid
        fb = _Fold(sd, p + ".bottom_dw_conv.bn", 1e-5)
        ddww = hip.dw3x3_wgrad(dsum, x_pre, Bi, e.Lv, g)                         # [9, D]
        fb.grads(grads, p + ".bottom_dw_conv.conv.weight", ddww.t().reshape(D, 1, 3, 3),
                 sd[p + ".bottom_dw_conv.conv.weight"].float(), cs)
        # pointwise conv: T = Wp (pool_out + shift)
        from .train import _wgrad
        ft = _Fold(sd, p + ".top2bottom_dw_conv.bn", 1e-5)
        pool_out = w["pool"][j]
        wp = sd[p + ".top2bottom_pw_conv.conv.weight"].float()[:, :, 0, 0]        # [D, C]
        dwp = _wgrad(dT_bf, pool_out, Bi * g2) + torch.outer(cs, ft.shift)
        grads[p + ".top2bottom_pw_conv.conv.weight"] = dwp.reshape(D, C, 1, 1)
        dshift = wp.t() @ cs                                                      # [C]
        wt = self._wt.get(("pw", j))
        if wt is None:
            wt = self._wt[("pw", j)] = a["pw"].weight[:, :C].t().contiguous()     # [C, D] bf16
        dpool = _zbuf(Bi * g2, C, dsum.device)
        hip.gemm(dT_bf, wt, dpool)
        # depthwise kernel == stride conv + BN
        dpf = hip.dwpool_wgrad(dpool, w["par"][j], Bi, hw, hw, C, k)              # [k*k, C]
        ft.grads(grads, p + ".top2bottom_dw_conv.conv.weight", dpf.t().reshape(C, 1, k, k),
                 sd[p + ".top2bottom_dw_conv.conv.weight"].float(), dshift)
        if self.dpar[0] is None:
            self.dpar[0] = _zbuf(Bi * hw * hw, C, dsum.device)
            hip.dwpool_bwd(dpool, a["pool"], self.dpar[0], Bi, hw, hw, C, k)
        else:
            hip.dwpool_bwd(dpool, a["pool"], self.dpar[0], Bi, hw, hw, C, k, accumulate=True)
        self._stage(grads, j)

    def _stage(self, grads, j):
        e, w, Bi, sd = self.e, self.w, self.Bi, self.sd
        da, db_ = self.dpar
        out = w["par"][j]
        dpre = self._relu_bwd(da, out, dy2=db_)
        self.dpar = [None, None]
        if j == 0:
            self._first_conv(grads, "visual.transformer.parallel_branch_v.0.conv.weight",
                             "visual.transformer.parallel_branch_v.0.bn", dpre)
            return
        q = f"visual.transformer.parallel_branch_v.{j}.resnet_stage.conv_0"
        c1, c2, cr, c3 = e.par_specs[j]
        t1, t2, _ = w["par_tmp"][j]
        src = w["par"][j - 1]

        def fold(conv, bn, G, dbias):
            _Fold(sd, f"{q}.{bn}", 1e-6).grads(grads, f"{q}.{conv}.weight", G, sd[f"{q}.{conv}.weight"].float(), dbias)
        G, dbias, dt2 = self._conv_bwd(("par", j, 3), c3, t2, dpre, Bi)
        fold("conv3", "bn3", G, dbias)
        G, dbias, dsrc_a = self._conv_bwd(("par", j, "r"), cr, src, dpre, Bi)
        fold("residual_conv", "residual_bn", G, dbias)
        del dpre
        dt2 = self._relu_bwd(dt2, t2)
        G, dbias, dt1 = self._conv_bwd(("par", j, 2), c2, t1, dt2, Bi)
        fold("conv2", "bn2", G, dbias)
        del dt2
        dt1 = self._relu_bwd(dt1, t1)
        G, dbias, dsrc_b = self._conv_bwd(("par", j, 1), c1, src, dt1, Bi)
        fold("conv1", "bn1", G, dbias)
        self.dpar = [dsrc_a, dsrc_b]

    # ------------------------------------------------------------------ stem
    def stem(self, grads, dtok):
        """dtok: fp32 [Mv, D] gradient of the token matrix in front of ln_pre (cls row included)."""
        from .train import _wgrad
        e, w, Bi, sd = self.e, self.w, self.Bi, self.sd
        g2, D = e.g * e.g, e.D
        sp = "visual.transformer.resblocks.0"
        dlast = hip.cast_bf16(dtok.view(Bi, e.Lv, D)[:, 1:].reshape(Bi * g2, D))
        x = w["stem"][-1]
        grads[sp + ".last_conv.weight"] = _wgrad(dlast, x[:Bi * g2], Bi * g2).reshape(D, x.shape[1], 1, 1)
        wt = self._wt.get("last")
        if wt is None:
            wt = self._wt["last"] = e.w_last.t().contiguous()                    # [Cin, D]
        dy = _zbuf(Bi * g2, x.shape[1], dtok.device)
        hip.gemm(dlast, wt, dy)
        for i in reversed(range(len(e.stem_specs))):
            spec = e.stem_specs[i]
            q = f"{sp}.resnet_stage.conv_{i}"
            x_in = w["stem"][i - 1] if i else w["S1"]
            dpre = self._relu_bwd(dy, w["stem"][i])
            G, dbias, dy = self._conv_bwd(("stem", i), spec, x_in, dpre, Bi)
            _Fold(sd, q + ".bn1", 1e-5).grads(grads, q + ".conv1.weight", G, sd[q + ".conv1.weight"].float(), dbias)
            _Fold(sd, q + ".downsample.1", 1e-5).grads(grads, q + ".downsample.0.weight", G[:, :, 1:2, 1:2].contiguous(),
                                                        sd[q + ".downsample.0.weight"].float(), dbias)
        dpre = self._relu_bwd(dy, w["S1"])
        self._first_conv(grads, sp + ".conv1.weight", sp + ".bn1", dpre)
        self.col_img = None
