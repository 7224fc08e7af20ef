"""Backward pass of the convolutional side of the vision tower (second slice): the EarlyconvRes
stem (M.py:1898-2000), the parallel convolutional branch (M.py:1812-1895) and the convolutional halves of the lateral
adapters (M.py:1752-1778).

Two BatchNorm semantics.  ConvSideBackward (first half of this file): FROZEN statistics -- the HIP forward folds every
BatchNorm's running mean / variance into the adjacent convolution (packing.py) and the backward differentiates exactly
that function: gamma, beta and the convolution weights receive gradients, the running statistics are constants (what
autograd of the reference gives in eval() mode; fixture tests/golden/*.grads.npz).  ConvSideBatchNorm (second half):
train-mode BatchNorm with per-GPU batch statistics, their backward and the running-statistics update (the reference in
train() mode; fixture tests/golden/b32-yfcc-msclips.grads_trainbn.npz).

Every convolution is a GEMM in the forward (implicit im2col gather); here
    dWf = dY^T . im2col(X)      msclip_im2col + the split-K wgrad GEMM (gradgemm.wgrad)
    dX  = col2im(dY . Wf)       msclip_gemm + msclip_col2im (pointwise convs: the GEMM writes dX directly)
    dbf = column sums of dY
on the FOLDED weight Wf = W * s, bf = beta - mean * s, s = gamma / sqrt(var + eps); the chain rule back to the module's
own parameters is a few per-channel vector operations (fold_grads).  The stem's 1x1 shortcut lives in the centre tap of
its 3x3 (packing.stem_stage): its gradient is the centre tap of the merged filter's gradient.

Activations are read from the engine's workspace (the training forward runs the conv side layer by layer with
engine.force_unfused, so every intermediate map is there): valid until the next forward on the same batch shape.
"""
import torch

from . import gradgemm
from . import hip
from . import options
from .gradgemm import wgrad as _wgrad, wgrad_async as _wgrad_async

BF = torch.bfloat16
F32 = torch.float32


def _zbuf(rows, cols, dev):
    """bf16 [rows, cols] with 64 zero elements of slack behind it (GEMM operands whose K is padded to 64 read past a
    row's end against zero weights: the bytes must be finite)."""
    n = rows * cols
    flat = torch.empty(n + 64, dtype=BF, device=dev)
    flat[n:].zero_()                                     # only the slack needs a defined value: the matrix itself is overwritten
    return flat[:n].view(rows, cols)


class _Fold:
    """One BatchNorm's frozen statistics: s = gamma * rstd, shift = beta - mean * s (formed lazily: only the adapters' pointwise
    conv needs `shift` on the host side of a launch)."""

    def __init__(self, sd, prefix, eps):
        self.prefix, self.eps = prefix, eps
        self.gamma, self.beta = sd[prefix + ".weight"], sd[prefix + ".bias"]
        self.mean, self.var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]

    @property
    def shift(self):
        return self.beta.float() - self.mean.float() * self.gamma.float() * torch.rsqrt(self.var.float() + self.eps)

    def grads(self, out, conv_key, G, w_raw, dshift):
        """G = dL/d(folded filter) in the raw filter's shape, dshift = dL/d(shift) -> raw conv / gamma / beta gradients
        (one launch: msclip_bn_fold_bwd)."""
        Gc = G if G.is_contiguous() else G.contiguous()
        dW, dg, db = hip.bn_fold_bwd(Gc, w_raw.contiguous(), dshift.contiguous(), self.gamma, self.mean, self.var, self.eps)
        out[conv_key] = dW
        out[self.prefix + ".weight"] = dg
        out[self.prefix + ".bias"] = db


_PARITY_PLANS = {}


def _image_kalign():
    return 32


class ConvSideBackward:
    def __init__(self, train_step):
        self.ts = train_step
        self.e = train_step.eng
        self._wt = {}
        self._pplan = _PARITY_PLANS                      # parity-class dgrad plans (geometry only: survive weight updates)

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

    # ---- input gradient of the 3x3 / stride 2 / pad 1 convolutions without column matrices (round 4).  An input pixel
    # (2i + py, 2j + px) receives from output pixel (i + a, j + b), a <= py, b <= px, through filter tap
    # ky = 1 (py = 0) or (2, 0)[a] (py = 1), kx likewise: each parity class is a small stride-1 convolution of dY
    # (1, 2, 2 or 4 taps -- 9 taps per 4 input pixels in total, the minimum), i.e. an implicit-GEMM launch
    # (msclip_gemm mode 1 over the NHWC dY) whose output rows scatter to the class's pixels: dX viewed as rows of two pixels
    # ([B * H * Wo, 2 * ci]), row m + (m / Wo) * Wo + py * Wo, column window px * ci (msclip_gemm_desc.rpg / radd / roff).
    # Replaces dcol = dY . Wf^T (a [pixels, 9 * ci] matrix written and read back: 1.4 GB for each of the two 112 x 112
    # layers at batch 512) + msclip_col2im: 1069 -> 314 us (48 -> 48 @ 112), 1463 -> 524 (48 -> 96 @ 112), 669 -> 252 (96 -> 96 @ 56),
    # 820 -> 367 (96 -> 192 @ 56), 367 -> 192 (192 -> 192 @ 28); closer to fp32 torch than the bf16 column matrix was
    # (tools/probes/dgrad_parity_probe.py).  On the large maps the two pixels of a pair share a launch (_parity_plan).
    def _parity_plan(self, key, spec):
        # geometry only: shared by every ConvSideBackward of the process (the frozen-statistics backward builds one per step,
        # and an index upload is a pageable host-to-device copy that waits for the stream to drain)
        # row pairs on the large maps (112 / 56: 543 -> 314 us, 758 -> 524, 419 -> 367), four classes on the 28 x 28 ones (192 vs 206 us)
        rows = spec.h_in >= 56
        key = (str(spec.weight.device), spec.cout, spec.cin, spec.weight.shape[1], spec.w_out, rows)
        plan = self._pplan.get(key)
        if plan is None:
            co, ci, kp = spec.cout, spec.cin, spec.weight.shape[1]
            src = torch.arange(co * kp).view(co, kp)[:, :9 * ci].view(co, 3, 3, ci)
            idx, parts, ktabs = [], [], []

            def add(t, py, px, ntaps_x, N):
                K = t.shape[1]
                Kp = (K + 63) // 64 * 64
                if Kp != K:
                    t = torch.cat([t, torch.full((t.shape[0], Kp - K), -1, dtype=t.dtype)], 1)
                parts.append((py, px, sum(x.numel() for x in idx), Kp, N))
                idx.append(t.reshape(-1))
                ktabs.append((1 + py, ntaps_x))
            for py in (0, 1):
                kys = [1] if py == 0 else [2, 0]
                if rows:
                    # ONE launch per input-row parity: both pixels of a pair (px = 0, 1) are 2 * ci output columns of the
                    # same GEMM row over the taps b = 0, 1 (px = 0 has no b = 1 tap: zero filter block) -- dY is read twice
                    # instead of four times and an output row is a full 2 * ci * 2-byte segment of dX
                    t0 = src[:, kys][:, :, [1]]                                            # px = 0: tap b = 0 is kx = 1
                    t0 = torch.cat([t0, torch.full_like(t0, -1)], 2)                       #         tap b = 1: nothing
                    t1 = src[:, kys][:, :, [2, 0]]                                         # px = 1: b = 0 -> kx = 2, b = 1 -> kx = 0
                    t = torch.cat([t0.permute(3, 1, 2, 0).reshape(ci, -1), t1.permute(3, 1, 2, 0).reshape(ci, -1)], 0)
                    add(t, py, 0, 2, 2 * ci)
                else:
                    for px in (0, 1):
                        kxs = [1] if px == 0 else [2, 0]
                        add(src[:, kys][:, :, kxs].permute(3, 1, 2, 0).reshape(ci, -1), py, px, 1 + px, ci)   # [ci, (a, b, co)]
            dev = spec.weight.device
            flat = torch.cat(idx)
            plan = self._pplan[key] = dict(idx=flat.clamp_min(0).to(dev), mask=(flat >= 0).to(torch.bfloat16).to(dev), parts=parts,
                                           ktab=[P.ktab_on(dev, kh_, kw_, co, spec.w_out) for kh_, kw_ in ktabs])
        return plan

    def _parity_ok(self, spec):
        return ((spec.kh, spec.kw, spec.stride, spec.pad) == (3, 3, 2, 1) and spec.h_in % 2 == 0 and spec.w_in % 2 == 0
                and spec.cout % 8 == 0 and spec.cin % 8 == 0 and spec.h_in == 2 * spec.h_out and spec.w_in == 2 * spec.w_out
                and spec.h_in >= 28                  # (14 x 14 maps: the four launches cost more than the small column matrix)
                and not options.TRAIN.dgrad_col2im)

    def _dgrad_parity(self, plan, spec, dpre, dx, B, relu_of=None):
        """relu_of: the saved post-ReLU activation the convolution read (bf16, laid out like dx): the launches then store
        dX * (activation > 0) -- the ReLU backward that follows in the chain, in the epilogue (msclip_gemm resid_kind 5)."""
        co, ci, Ho, Wo = spec.cout, spec.cin, spec.h_out, spec.w_out
        wp = torch.index_select(spec.weight.reshape(-1), 0, plan["idx"]).mul_(plan["mask"])      # the class filters: one gather, zero blocks masked
        rows = B * spec.h_in * Wo
        dx2 = dx.view(rows, 2 * ci)
        y2 = relu_of[:B * spec.h_in * spec.w_in].view(rows, 2 * ci) if relu_of is not None else None
        for (py, px, off, Kp, N), kt in zip(plan["parts"], plan["ktab"]):
            hip.gemm(dpre, wp[off:off + N * Kp].view(N, Kp), dx2[:, px * ci:px * ci + N], M=B * Ho * Wo, N=N,
                     conv=(Ho, Wo, co, Ho, Wo, 1, 0), ktab=kt, ldo=2 * ci, rpg=Wo, radd=Wo, roff=py * Wo,
                     resid=y2[:, px * ci:px * ci + N] if y2 is not None else None, ldr=2 * ci,
                     resid_kind=hip.RESID_RELUMASK if y2 is not None else 0)

    def _shortcut_ok(self, spec):
        return ((spec.kh, spec.kw, spec.stride, spec.pad) == (1, 1, 2, 0) and spec.h_in == 2 * spec.h_out
                and spec.w_in == 2 * spec.w_out and spec.cin % 8 == 0 and not options.TRAIN.dgrad_col2im)

    def _shortcut_dgrad_into(self, key, spec, dpre, dx, B):
        """dx[even pixels] += dY . W for a 1x1 / stride-2 convolution (the bottleneck's shortcut), in place into the input
        gradient `dx` another path has already written: one GEMM whose rows scatter to the pixels (2i, 2j) and add what is there
        (msclip_gemm resid_kind 6) -- no column matrix, no col2im into a zero-filled map, no second operand for the ReLU pass."""
        co, ci, Ho, Wo = spec.cout, spec.cin, spec.h_out, spec.w_out
        wt = self._w_t(key, spec.weight, co, ci)
        dx2 = dx.view(B * spec.h_in * Wo, 2 * ci)[:, :ci]
        hip.gemm(dpre, wt, dx2, M=B * Ho * Wo, N=ci, ldx=co, ldo=2 * ci, rpg=Wo, radd=Wo, roff=0,
                 resid=dx2, ldr=2 * ci, resid_kind=hip.RESID_ACCUM)

    def _conv_bwd(self, key, spec, x_in, dpre, B, need_dx=True, col=None, lane=False, relu_of=None):
        """dpre: bf16 [B*Ho*Wo, cout] (with slack) -> (G [cout, cin, kh, kw] fp32 wrt the folded filter, dbias [cout],
        dx NHWC bf16 [B*H*W, cin] or None).  lane=True: the weight gradient runs on gradgemm's lane stream (G may only be
        read after gradgemm.join) and no bias gradient is formed.  relu_of: dx is returned through the backward of the ReLU
        that produced this (bf16) activation -- in the parity launches' epilogues where they run, a msclip_relu_bwd pass else."""
        co, ci, kh, kw = spec.cout, spec.cin, spec.kh, spec.kw
        pix = B * spec.h_out * spec.w_out
        K = kh * kw * ci
        pointwise = (kh, kw, spec.stride, spec.pad) == (1, 1, 1, 0)
        if col is None:
            if pointwise:
                col = x_in[:pix]
            elif lane and not options.TRAIN.im2col_main:
                # only the weight gradient reads the column matrix: it is built where that runs, on the lane stream, not
                # in front of the input gradient on the critical path (x_in is a workspace map: nothing writes it before
                # gradgemm.join at the end of the backward)
                def col():
                    return hip.im2col(x_in, B, spec.h_in, spec.w_in, ci, kh, kw, spec.stride, spec.pad)
            else:
                col = hip.im2col(x_in, B, spec.h_in, spec.w_in, ci, kh, kw, spec.stride, spec.pad)

        def to_filter(dwf):
            return dwf[:, :K].reshape(co, kh, kw, ci).permute(0, 3, 1, 2).contiguous()
        if lane:
            G, db = _wgrad_async(dpre, col, pix, post=to_filter), None
        else:
            G, db = to_filter(_wgrad(dpre, col, pix)), hip.colsum(dpre, M=pix)
        dx = None
        if need_dx:
            if pointwise:
                wt = self._w_t(key, spec.weight, co, ci)
                dx = _zbuf(pix, ci, dpre.device)
                hip.gemm(dpre, wt, dx, M=pix, N=ci, ldx=co)
            elif self._parity_ok(spec) and self._parity_plan(key, spec):
                dx = _zbuf(B * spec.h_in * spec.w_in, ci, dpre.device)
                fuse = relu_of is not None and relu_of.dtype == BF
                self._dgrad_parity(self._parity_plan(key, spec), spec, dpre, dx, B, relu_of=relu_of if fuse else None)
                if fuse:
                    relu_of = None
            else:
                kp = spec.weight.shape[1]
                wt = self._w_t(key, spec.weight, co, kp)
                dcol = torch.empty(pix, kp, dtype=BF, device=dpre.device)
                hip.gemm(dpre, wt, dcol, M=pix, N=kp, ldx=co)
                dx = _zbuf(B * spec.h_in * spec.w_in, ci, dpre.device)
                hip.col2im(dcol, dx, B, spec.h_in, spec.w_in, ci, kh, kw, spec.stride, spec.pad)
                del dcol
            if relu_of is not None:
                dx = self._relu_bwd(dx, relu_of)
        return G, db, dx

    def _fold_on_lane(self, grads, fold, conv_key, G, w_raw, dshift, also=None):
        """fold.grads(...) on gradgemm's lane stream, behind the weight gradient G it consumes (G comes from
        _conv_bwd(lane=True)); `also` = (fold2, key2, w_raw2) reads the centre tap of the same G (the stem's merged 1x1
        shortcut).  The results may only be touched after gradgemm.join, like every lane product."""
        def fn():
            fold.grads(grads, conv_key, G, w_raw, dshift)
            if also is not None:
                f2, k2, w2 = also
                f2.grads(grads, k2, G[:, :, 1:2, 1:2].contiguous(), w2, dshift)
            return grads[conv_key]
        gradgemm.on_lane(fn, G, dshift)
        if G.is_cuda:
            cur = torch.cuda.current_stream(G.device)
            for f, k in ((fold, conv_key),) + (((also[0], also[1]),) if also is not None else ()):
                for key in (k, f.prefix + ".weight", f.prefix + ".bias"):
                    grads[key].record_stream(cur)

    def _colsum_on_lane(self, dpre, M):
        """Bias gradient (column sums of dY) for a fold whose chain rule runs on the lane anyway (_fold_on_lane): the pass over
        dY moves off the critical path with it (options.TRAIN.colsum_main: on the calling stream, as before)."""
        if options.TRAIN.colsum_main:
            return hip.colsum(dpre, M=M)
        return gradgemm.on_lane(lambda: hip.colsum(dpre, M=M), dpre)

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

    def _image_cols(self, gemm_operand=False):
        """The image's 3 x 3 / stride 2 patch matrix (both Cin = 3 convolutions share it).  gemm_operand: it will be the X operand
        of a forward GEMM (train-mode BatchNorm's raw convolutions), whose K axis is 64 wide; whoever asks first decides."""
        if self.col_img is None:
            S = self.e.S
            # 27 columns padded to 32, not 64: the token-major wgrad GEMM reads it as it is (half the bytes, written and read twice)
            self.col_img = hip.im2col(self.img, self.Bi, S, S, 3, 3, 3, 2, 1, image=True, kalign=64 if gemm_operand else _image_kalign())
        return self.col_img

    def _first_conv(self, grads, conv_key, bn_prefix, dpre):
        """wgrad of a 3x3 / stride 2 convolution on the input image (no input gradient)."""
        pix = self.Bi * self.e.h1 * self.e.h1
        co = dpre.shape[1]
        if hip.image_conv_wgrad_ok(self.img, dpre) and self.e.h1 <= 128:
            # one pass over dpre and the image: no patch matrix, the bias sums in the same contraction (round 5)
            dwf, dbias = hip.image_conv_wgrad(self.img, dpre)
        else:
            dwf, dbias = _wgrad(dpre, self._image_cols(), pix)[:, :27], hip.colsum(dpre, M=pix)
        G = dwf.reshape(co, 3, 3, 3).permute(0, 3, 1, 2).contiguous()          # (kh, kw, ci) -> [co, ci, kh, kw]
        _Fold(self.sd, bn_prefix, 1e-5).grads(grads, conv_key, G, self.sd[conv_key].float(), dbias)

    # ------------------------------------------------------------------ lateral adapter j + parallel stage j
    def adapter(self, grads, j, dsum, x_pre):
        """dsum: fp32 [Mv, D] gradient of the adapter's pre-LayerNorm sum; x_pre: the token matrix the adapter read."""
        e, w, Bi, sd = self.e, self.w, self.Bi, self.sd
        a = e.adapters[j]
        g, D, C, k, hw = e.g, e.D, a["C"], a["k"], e.par_hw[j]
        g2 = g * g
        p = f"visual.transformer.parallel_lateral_adapter.{j}"
        # grid rows (the cls row has no top-down term): their bf16 copy and column sums from ONE pass over dsum (round 5; was a
        # gathering copy, a column-sum pass and a cast)
        if dsum.is_contiguous() and dsum.shape[0] == Bi * e.Lv and e.Lv == g2 + 1:
            dT_bf, cs = hip.cast_bf16_colsum(dsum, skip_group=g2)
        else:
            dT = dsum.view(Bi, e.Lv, D)[:, 1:].reshape(Bi * g2, D)
            cs = hip.colsum(dT)
            dT_bf = hip.cast_bf16(dT)
        # bottom depthwise 3x3 + BN on the token grid
        fb = _Fold(sd, p + ".bottom_dw_conv.bn", 1e-5)
        ddww = hip.dw3x3_wgrad(dsum, x_pre, Bi, e.Lv, g)                         # [9, D]
        fb.grads(grads, p + ".bottom_dw_conv.conv.weight", ddww.t().reshape(D, 1, 3, 3),
                 sd[p + ".bottom_dw_conv.conv.weight"].float(), cs)
        # pointwise conv: T = Wp (pool_out + shift)
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

        def fold(conv, bn, G, dpre_, spec):
            # weight gradient + the fold's chain rule on the lane stream; the bias sum (main stream) feeds the latter
            dbias = self._colsum_on_lane(dpre_, Bi * spec.h_out * spec.w_out)
            self._fold_on_lane(grads, _Fold(sd, f"{q}.{bn}", 1e-6), f"{q}.{conv}.weight", G, sd[f"{q}.{conv}.weight"].float(), dbias)
        G, _, dt2 = self._conv_bwd(("par", j, 3), c3, t2, dpre, Bi, lane=True)
        fold("conv3", "bn3", G, dpre, c3)
        short = self._shortcut_ok(cr)        # the shortcut's input gradient is added into conv1's below (no map of its own)
        G, _, dsrc_a = self._conv_bwd(("par", j, "r"), cr, src, dpre, Bi, lane=True, need_dx=not short)
        fold("residual_conv", "residual_bn", G, dpre, cr)
        if not short:
            del dpre
        dt2 = self._relu_bwd(dt2, t2)
        G, _, dt1 = self._conv_bwd(("par", j, 2), c2, t1, dt2, Bi, lane=True, relu_of=t1)      # (through t1's ReLU)
        fold("conv2", "bn2", G, dt2, c2)
        del dt2
        G, _, dsrc_b = self._conv_bwd(("par", j, 1), c1, src, dt1, Bi, lane=True)
        fold("conv1", "bn1", G, dt1, c1)
        if short:
            self._shortcut_dgrad_into(("par", j, "r"), cr, dpre, dsrc_b, Bi)
            dsrc_a, dsrc_b = dsrc_b, None
        self.dpar = [dsrc_a, dsrc_b]

    # ------------------------------------------------------------------ stem
    def stem(self, grads, dtok):
        """dtok: fp32 [Mv, D] gradient of the token matrix in front of ln_pre (cls row included)."""
        e, w, Bi, sd = self.e, self.w, self.Bi, self.sd
        g2, D = e.g * e.g, e.D
        sp = "visual.transformer.resblocks.0"
        dlast = hip.cast_bf16_colsum(dtok[:Bi * e.Lv], fold=False, skip_group=g2)[0]     # the grid rows (class rows skipped), one pass, no gather copy
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
            # dy arrives through the ReLU of this stage's output: from the previous iteration's convolution backward
            # (relu_of = its input map = this stage's output), or the pass below for the last stage
            dpre = self._relu_bwd(dy, w["stem"][i]) if i == len(e.stem_specs) - 1 else dy
            G, _, dy = self._conv_bwd(("stem", i), spec, x_in, dpre, Bi, lane=True, relu_of=x_in)
            dbias = self._colsum_on_lane(dpre, Bi * spec.h_out * spec.w_out)
            self._fold_on_lane(grads, _Fold(sd, q + ".bn1", 1e-5), q + ".conv1.weight", G, sd[q + ".conv1.weight"].float(), dbias,
                               also=(_Fold(sd, q + ".downsample.1", 1e-5), q + ".downsample.0.weight",
                                     sd[q + ".downsample.0.weight"].float()))
        self._first_conv(grads, sp + ".conv1.weight", sp + ".bn1", dy)          # (dy already went through S1's ReLU: relu_of above)
        self.col_img = None


# ----------------------------------------------------------------------------------------------------------------------
# Train-mode BatchNorm (per-GPU batch statistics): what nn.BatchNorm2d does in train() (M.py:1825-1861, 1920-1936).
# Forward: every convolution runs with its RAW weights (no fold), the batch mean / biased variance of its output are
# reduced on the GPU (msclip_bn_stats), normalise + affine (+ residual, ReLU) is one element-wise pass (msclip_bn_apply),
# and the running statistics get the reference's update (momentum 0.1, unbiased variance).  Backward: dgamma / dbeta and
# the gradient of the raw convolution output (msclip_bn_bwd_reduce / _dx), then the same GEMM-based convolution backward
# as above on the raw weights (no fold chain rule).  The stem's 3x3 and its 1x1 shortcut are separate convolutions here:
# their two BatchNorms see different statistics, the centre-tap merge only holds for frozen statistics.
# ----------------------------------------------------------------------------------------------------------------------
from . import packing as P


class _RawSpecs:
    """Raw-weight ConvSpecs (geometry and chunk tables built once; weights refreshed from the module every step)."""

    def __init__(self, e):
        self.e = e
        sd = e.state_views()
        dev = e.dev
        sp = "visual.transformer.resblocks.0"

        def spec(key, h, s, pad):
            wt = sd[key + ".weight"].float()
            return P.ConvSpec(wt, torch.zeros(wt.shape[0]), h, h, s, pad).to(dev), key + ".weight"
        self.items = []                                  # (spec, state_dict key)
        self.stem = []
        for i, fs in enumerate(e.stem_specs):
            q = f"{sp}.resnet_stage.conv_{i}"
            main, short = spec(q + ".conv1", fs.h_in, fs.stride, 1), spec(q + ".downsample.0", fs.h_in, fs.stride, 0)
            self.stem.append((main[0], short[0], q))
            self.items += [main, short]
        self.par = [None]
        for j in range(1, 5):
            q = f"visual.transformer.parallel_branch_v.{j}.resnet_stage.conv_0"
            c1f, c2f, crf, c3f = e.par_specs[j]
            quad = [spec(q + ".conv1", c1f.h_in, 1, 0), spec(q + ".conv2", c2f.h_in, c2f.stride, c2f.pad),
                    spec(q + ".residual_conv", crf.h_in, crf.stride, 0), spec(q + ".conv3", c3f.h_in, 1, 0)]
            self.par.append(tuple(x[0] for x in quad) + (q,))
            self.items += quad
        self.pw = []
        for j in range(5):
            p = f"visual.transformer.parallel_lateral_adapter.{j}"
            pw = spec(p + ".top2bottom_pw_conv.conv", e.g, 1, 0)
            self.pw.append(pw[0])
            self.items.append(pw)
        self.refresh()

    def refresh(self):
        """The raw operands of this step from the module's parameters.  options.TRAIN.raw_pack_table (default): ONE msclip_pack_weights
        launch from a device-resident item table (built once per set of parameter tensors) rewrites every one of them in place;
        else ~60 ATen permute / cast / copy launches (round 5; the cross-check of the table)."""
        e = self.e
        sd = e.state_views()
        if options.TRAIN.raw_pack_table and e.dev.type == "cuda":
            if getattr(self, "_table", None) is None or self._table_sd is not sd:
                self._refresh_eager(sd)                      # allocates the operand tensors (and fills them once)
                self._table, self._table_sd = self._build_table(sd), sd
            self._table.run()
            self.sd = sd
            return
        self._refresh_eager(sd)

    def _build_table(self, sd):
        t = hip.PackPlan(self.e.dev)
        for sp_, key in self.items:
            t.add(sd[key], out=sp_.weight)                   # bf16 [co][kpad], (kh, kw, ci) order, zero padded
        sp = "visual.transformer.resblocks.0"
        wa, wb = sd[sp + ".conv1.weight"], sd["visual.transformer.parallel_branch_v.0.conv.weight"]
        t.add(wa, out=self.w_conv1)
        t.add(wb, out=self.w_par0)
        if self.w_dual is not None:                          # fp32 [27][96]: rows (ci, kh, kw) as the filters lie in memory
            t.add(wa, out=self.w_dual, mode=1, col0=0, ld=96)
            t.add(wb, out=self.w_dual, mode=1, col0=48, ld=96)
        for j in range(5):
            p = f"visual.transformer.parallel_lateral_adapter.{j}"
            wd = sd[p + ".top2bottom_dw_conv.conv.weight"]
            t.add(wd, out=self.pool[j], mode=1, col0=0, ld=wd.shape[0])      # [k*k][C]
            wbt = sd[p + ".bottom_dw_conv.conv.weight"]
            t.add(wbt, out=self.dww[j], mode=1, col0=0, ld=wbt.shape[0])     # [9][D]
        return t.finalize()

    def _refresh_eager(self, sd):
        e = self.e
        for sp_, key in self.items:
            # [Cout, Cin, KH, KW] fp32 -> the spec's [Cout, Kpad] bf16 matrix in (kh, kw, ci) order: ONE strided, converting copy
            # into its first K columns (the pad columns stay zero) instead of permute / pad / cast / copy
            wt = sd[key]
            co, ci, kh, kw = wt.shape
            sp_.weight[:, :kh * kw * ci].view(co, kh, kw, ci).copy_(wt.permute(0, 2, 3, 1))
        sp = "visual.transformer.resblocks.0"

        def first(key):                                  # [48, 3, 3, 3] -> [48, 64] bf16 in the patch matrix's (kh, kw, ci) order
            return P.pad_k(P.conv_weight_matrix(sd[key].float())).to(BF).contiguous()
        self.w_conv1 = first(sp + ".conv1.weight")
        self.w_par0 = first("visual.transformer.parallel_branch_v.0.conv.weight")
        # both filters as the fused pass reads them: fp32 [27][96], row ci * 9 + kh * 3 + kw (msclip_stem_conv3x3s2_dual_raw)
        wa, wb = sd[sp + ".conv1.weight"], sd["visual.transformer.parallel_branch_v.0.conv.weight"]
        self.w_dual = (torch.cat([wa.reshape(wa.shape[0], 27), wb.reshape(wb.shape[0], 27)], 0).t().float().contiguous()
                       if wa.shape[0] == 48 and wb.shape[0] == 48 else None)
        self.pool, self.dww = [], []
        for j in range(5):
            p = f"visual.transformer.parallel_lateral_adapter.{j}"
            wd = sd[p + ".top2bottom_dw_conv.conv.weight"].float()
            c, _, k, _ = wd.shape
            self.pool.append(wd[:, 0].reshape(c, k * k).t().contiguous())                      # [k*k, C]
            wb = sd[p + ".bottom_dw_conv.conv.weight"].float()
            self.dww.append(wb[:, 0].reshape(wb.shape[0], 9).t().contiguous())                 # [9, D]
        self.sd = sd


class ConvSideBatchNorm:
    """Forward and backward of the conv side with train-mode BatchNorm.  One instance per TrainStep."""
    MOMENTUM = 0.1

    def __init__(self, train_step):
        self.ts = train_step
        self.e = train_step.eng
        self.raw = _RawSpecs(self.e)
        self._two_pass = {}                              # conv shape -> does msclip_gemm take it in its BatchNorm modes
        self._zeros = {}                                 # read-only zero operands of the adapters' raw depthwise pass
        self.bw = ConvSideBackward(train_step)           # its generic pieces (_conv_bwd, _relu_bwd, patch matrix) are reused

    # ------------------------------------------------------------------ forward
    def _bn(self, x_raw, prefix, eps, out, M, relu=False, resid=None):
        sd = self.raw.sd
        g = sd[prefix + ".weight"]
        mean, var, rstd, scale, shift = hip.bn_stats(x_raw, M, gamma=g, beta=sd[prefix + ".bias"], eps=eps)
        hip.bn_apply(x_raw, scale, shift, out, M, relu=relu, resid=resid)
        self.saved[prefix] = (x_raw, mean, rstd, g, M)
        self.stats[prefix] = (mean, var, M)
        return out

    def _two_pass_ok(self, spec, M):
        """Does msclip_gemm run this convolution in its BatchNorm modes (the streaming kernel's shapes)?  Asked once per shape."""
        if not options.TRAIN.bn_two_pass:
            return False
        pointwise = spec.kh == 1 and spec.kw == 1 and spec.stride == 1 and spec.pad == 0
        key = (spec.cin, spec.cout, spec.weight.shape[1], M, None if pointwise else spec.geometry())
        if key not in self._two_pass:
            self._two_pass[key] = hip.gemm_bn_two_pass_ok(spec.cin, spec.cout, spec.weight.shape[1], M, conv=key[4])
        return self._two_pass[key]

    def _conv_bn(self, x, spec, prefix, eps, out, M, relu=False, resid=None, two_pass=None):
        """out = [relu](BN_batch(conv(x)) [+ resid]).  Convolutions the streaming kernel runs (pointwise, 1 x 1 stride 2, 3 x 3 over 48
        channels) in two passes over x -- statistics, then normalise in the epilogue, xhat (bf16) kept for the backward; no raw map
        (options.TRAIN.bn_two_pass) --, the others as raw fp32 map + statistics pass + normalise pass.  two_pass: the caller's choice
        for BatchNorms whose backward shares one pass (both sides of a residual block must keep the same kind of map)."""
        e, Bi = self.e, self.Bi
        pointwise = spec.kh == 1 and spec.kw == 1 and spec.stride == 1 and spec.pad == 0
        kw = dict(M=M, N=spec.cout, ldx=spec.cin) if pointwise else dict(M=M, N=spec.cout, conv=spec.geometry(), ktab=spec.ktab)
        if two_pass is None:
            two_pass = self._two_pass_ok(spec, M)
        if two_pass:                                         # (raw specs carry a zero bias)
            assert out.stride(0) == spec.cout and (resid is None or resid.stride(0) == spec.cout)
            part = self._part_rows(spec.cout, e.dev)
            hip.gemm(x, spec.weight, part, bn_stats_part=part, **kw)
            sd = self.raw.sd
            g = sd[prefix + ".weight"]
            sums = hip.colsum(part)
            o = torch.empty(5, spec.cout, dtype=F32, device=e.dev)
            hip.bn_finish(sums, spec.cout, M, g, sd[prefix + ".bias"], eps, o)
            mean, var, rstd, scale, shift = (o[k] for k in range(5))
            xh = _zbuf(out.shape[0], spec.cout, e.dev)
            hip.gemm(x, spec.weight, out, out2=xh, bn_consts=o, act=hip.ACT_RELU if relu else hip.ACT_NONE, resid=resid,
                     resid_kind=hip.RESID_BF16 if resid is not None else hip.RESID_NONE, **kw)
            zero, one = hip._bn_unit(spec.cout, e.dev)[1], hip._bn_unit(spec.cout, e.dev)[0]
            self.saved[prefix] = (xh, zero, one, scale, M)
            self.stats[prefix] = (mean, var, M)
            return out
        r = torch.empty(M, spec.cout, dtype=F32, device=e.dev)       # raw conv outputs stay fp32 until normalised
        e._conv(x, spec, r, Bi)
        return self._bn(r, prefix, eps, out, M, relu=relu, resid=resid)

    def _part_rows(self, N, dev):
        """Zeroed [2048, 2 N] partial-sum rows of a statistics launch (a wave per row; rows of waves that do not exist stay zero)."""
        return torch.zeros(2048, 2 * N, dtype=F32, device=dev)

    def begin(self, img, w, Bi):
        self.raw.refresh()
        self.saved, self.stats = {}, {}
        self.w, self.Bi, self.img = w, Bi, img
        self.bw.begin(img, w, Bi)
        self.bw.sd = self.raw.sd
        self.bw._wt = {}                                 # transposed dgrad weights of the previous step are stale

    def front(self):
        """conv1 / parallel stage 0 (both from one patch matrix of the image) and the four stem stages -> w["S1"], w["P0"],
        w["stem"][i]."""
        e, w, Bi = self.e, self.w, self.Bi
        sp = "visual.transformer.resblocks.0"
        pix = Bi * e.h1 * e.h1
        heads = ((self.raw.w_conv1, sp + ".bn1", e._s1(w, Bi)), (self.raw.w_par0, "visual.transformer.parallel_branch_v.0.bn", w["P0"]))
        if self.raw.w_dual is not None and e.S % 2 == 0 and options.TRAIN.bn_two_pass:
            # two passes over the image (round 6): statistics, then convolution + normalise + ReLU; no raw map exists, the backward
            # reads the normalised values xhat (bf16) with mean 0 / rstd 1 / gamma := gamma rstd
            sd = self.raw.sd
            xh = [_zbuf(pix, 48, e.dev) for _ in heads]
            res = hip.stem_conv_dual_bn(self.img, self.raw.w_dual, [(sd[p + ".weight"], sd[p + ".bias"]) for _, p, _ in heads],
                                        heads[0][2], heads[1][2], xh[0], xh[1], eps=1e-5)
            zero, one = hip._bn_unit(48, e.dev)[1], hip._bn_unit(48, e.dev)[0]
            for (_, prefix, _), x, (mean, var, rstd, scale, shift) in zip(heads, xh, res):
                self.saved[prefix] = (x, zero, one, scale, pix)
                self.stats[prefix] = (mean, var, pix)
            raws = []
            heads = ()
        elif self.raw.w_dual is not None and e.S % 2 == 0:
            # both raw convolutions from ONE pass over the image (round 5): no patch matrix, no two GEMMs over it
            raws = [torch.empty(pix, 48, dtype=F32, device=e.dev) for _ in heads]
            hip.stem_conv_dual_raw(self.img, self.raw.w_dual, raws[0], raws[1])
        else:
            col = self.bw._image_cols(gemm_operand=True)
            raws = []
            for wt, _, _ in heads:
                r = torch.empty(pix, wt.shape[0], dtype=F32, device=e.dev)  # raw conv outputs stay fp32 until normalised
                hip.gemm(col, wt, r)
                raws.append(r)
        for (wt, prefix, out), r in zip(heads, raws):
            self._bn(r, prefix, 1e-5, out, pix, relu=True)
        x = w["S1"]
        for i, (main, short, q) in enumerate(self.raw.stem):
            M = Bi * main.h_out * main.w_out
            tmp = _zbuf(M, main.cout, e.dev)
            tp = self._two_pass_ok(main, M) and self._two_pass_ok(short, M)
            self._conv_bn(x, main, q + ".bn1", 1e-5, tmp, M, two_pass=tp)
            self._conv_bn(x, short, q + ".downsample.1", 1e-5, w["stem"][i], M, relu=True, resid=tmp, two_pass=tp)
            x = w["stem"][i]

    def stage(self, j):
        if j == 0:
            return
        e, w, Bi = self.e, self.w, self.Bi
        c1, c2, cr, c3, q = self.raw.par[j]
        t1, t2, tr = w["par_tmp"][j]
        src = w["par"][j - 1]

        def run(spec, x, bn, out, relu, resid=None, two_pass=None):
            M = Bi * spec.h_out * spec.w_out
            self._conv_bn(x, spec, f"{q}.{bn}", 1e-6, out, M, relu=relu, resid=resid, two_pass=two_pass)
        run(c1, src, "bn1", t1, True)
        run(c2, t1, "bn2", t2, True)
        M3 = Bi * c3.h_out * c3.w_out
        tp = self._two_pass_ok(cr, M3) and self._two_pass_ok(c3, M3)    # bn3 and residual_bn share their backward pass
        run(cr, src, "residual_bn", tr, False, two_pass=tp)
        run(c3, t2, "bn3", w["par"][j], True, resid=tr, two_pass=tp)

    def adapter_top(self, j, out):
        """pool = BN(dwconv_{k=s}(par[j])) -> w["pool"][j]; out = pw(pool)."""
        e, w, Bi = self.e, self.w, self.Bi
        a = e.adapters[j]
        hw, g2 = e.par_hw[j], e.g * e.g
        p = f"visual.transformer.parallel_lateral_adapter.{j}"
        r = _zbuf(Bi * g2, a["C"], e.dev)
        hip.dwpool(w["par"][j], self.raw.pool[j], r, Bi, hw, hw, a["C"], a["k"])
        self._bn(r, p + ".top2bottom_dw_conv.bn", 1e-5, w["pool"][j], Bi * g2)
        pw = self.raw.pw[j]
        hip.gemm(w["pool"][j], pw.weight, out, M=Bi * g2, N=pw.cout, ldx=pw.cin)

    def adapter_sum(self, j, X, t, out):
        """out = [2 cls; BN(dw3x3(grid)) + t] with batch statistics of the depthwise output over the grid rows."""
        e, Bi = self.e, self.Bi
        D, g, L = e.D, e.g, e.Lv
        p = f"visual.transformer.parallel_lateral_adapter.{j}.bottom_dw_conv.bn"
        sd = self.raw.sd
        gam = sd[p + ".weight"]
        raw_full = torch.empty(Bi * L, D, dtype=F32, device=e.dev)
        if options.TRAIN.adapter_bn_views:
            # no copy of the grid rows and no fill: a sample is ONE row of L * D columns, the statistics are taken per (token, channel)
            # column and the class token's columns are left out of the fold over tokens
            key = (Bi * g * g, D)
            if key not in self._zeros:
                self._zeros[key] = (torch.zeros(Bi * g * g, D, dtype=F32, device=e.dev), torch.zeros(D, dtype=F32, device=e.dev))
            zero_t, zero_b = self._zeros[key]                            # (read-only)
            hip.adapter_sum(X, zero_t, self.raw.dww[j], zero_b, raw_full, Bi, L, g, e.usecls)
            part = hip.bn_stats_partials(raw_full.view(Bi, L * D))       # [1][2][L * D]
            sums = part.view(2, L, D)[:, 1:].sum(1)
            o = torch.empty(5, D, dtype=F32, device=e.dev)
            hip.bn_finish(sums, D, Bi * g * g, gam, sd[p + ".bias"], 1e-5, o)
            mean, var, rstd, scale, shift = (o[k] for k in range(5))
            hip.adapter_sum(X, t, (self.raw.dww[j] * scale).contiguous(), shift, out, Bi, L, g, e.usecls)
            self.saved[p] = (raw_full, mean, rstd, gam, Bi * g * g, L)   # (6 entries: the map still holds the class rows)
            self.stats[p] = (mean, var, Bi * g * g)
            return
        zero_t = torch.zeros(Bi * g * g, D, dtype=F32, device=e.dev)
        zero_b = torch.zeros(D, dtype=F32, device=e.dev)
        hip.adapter_sum(X, zero_t, self.raw.dww[j], zero_b, raw_full, Bi, L, g, e.usecls)
        graw = raw_full.view(Bi, L, D)[:, 1:].reshape(Bi * g * g, D)
        mean, var, rstd, scale, shift = hip.bn_stats(graw, gamma=gam, beta=sd[p + ".bias"], eps=1e-5)
        hip.adapter_sum(X, t, (self.raw.dww[j] * scale).contiguous(), shift, out, Bi, L, g, e.usecls)
        self.saved[p] = (graw, mean, rstd, gam, Bi * g * g)
        self.stats[p] = (mean, var, Bi * g * g)

    def update_running_stats(self):
        """running = (1 - m) running + m batch (unbiased variance), num_batches_tracked += 1: in place on the module's buffers."""
        bufs = dict(self.e.model.named_buffers())
        m = self.MOMENTUM
        keys = list(self.stats)
        if not keys:
            return
        with torch.no_grad():                            # multi-tensor forms: a handful of launches for the 36 BatchNorms
            rm = [bufs[k + ".running_mean"] for k in keys]
            rv = [bufs[k + ".running_var"] for k in keys]
            torch._foreach_mul_(rm, 1 - m)
            torch._foreach_add_(rm, [self.stats[k][0] for k in keys], alpha=m)
            torch._foreach_mul_(rv, 1 - m)
            unbiased = torch._foreach_mul([self.stats[k][1] for k in keys], [m * n / max(n - 1, 1) for _, _, n in (self.stats[k] for k in keys)])
            torch._foreach_add_(rv, unbiased)
            torch._foreach_add_([bufs[k + ".num_batches_tracked"] for k in keys], 1)

    # ------------------------------------------------------------------ backward
    def _bn_bwd(self, grads, prefix, dy, _plain=False):
        if not _plain and options.TRAIN.bn_bwd_fused and dy.dtype == BF:
            return self._bn_bwd_masked(grads, [prefix], dy)[0]       # (the 4-columns-per-thread passes; falls back to here)
        x_raw, mean, rstd, gam, M = self.saved[prefix]
        dx = torch.empty_like(dy) if dy.dtype == F32 else _zbuf(x_raw.shape[0], x_raw.shape[1], x_raw.device)
        dg, db = hip.bn_bwd(dy, x_raw, mean, rstd, gam, dx, M)
        grads[prefix + ".weight"], grads[prefix + ".bias"] = dg, db
        return dx

    def _bn_bwd_masked(self, grads, prefixes, dy, y=None, dy2=None):
        """The BatchNorms `prefixes` (one, or a residual block's two) behind d = (dy [+ dy2]) * (y > 0): one fused reduce + one fused
        dx pass (msclip_bn_bwd_fused) where the shapes allow, the ReLU pass + a pass pair per BatchNorm else.  -> [dx per prefix]"""
        saved = [self.saved[p] for p in prefixes]
        M = saved[0][4]
        if options.TRAIN.bn_bwd_fused and dy.dtype == BF and all(sv[4] == M for sv in saved):
            dxs = [_zbuf(sv[0].shape[0], sv[0].shape[1], sv[0].device) for sv in saved]
            sides = [(sv[0], sv[1], sv[2], sv[3], dx) for sv, dx in zip(saved, dxs)]
        else:
            sides = None
        if sides is not None and hip.bn_bwd_fused_ok(dy, sides, y, dy2, M):
            res = hip.bn_bwd_fused(dy, sides, y=y, dy2=dy2, M=M)
            for p, (dg, db) in zip(prefixes, res):
                grads[p + ".weight"], grads[p + ".bias"] = dg, db
            return dxs
        dpre = self.bw._relu_bwd(dy, y, dy2=dy2) if y is not None else dy
        assert y is not None or dy2 is None
        return [self._bn_bwd(grads, p, dpre, _plain=True) for p in prefixes]

    def _conv(self, grads, key, spec, wkey, x_in, draw, need_dx=True, relu_of=None):
        G, _, dx = self.bw._conv_bwd(key, spec, x_in, draw, self.Bi, need_dx=need_dx, lane=True, relu_of=relu_of)
        grads[wkey] = G                                  # final after gradgemm.join (end of the backward / bucket flush)
        return dx

    def _first(self, grads, wkey, prefix, dy, y=None, dy2=None):
        draw, = self._bn_bwd_masked(grads, [prefix], dy, y=y, dy2=dy2)
        pix = self.Bi * self.e.h1 * self.e.h1
        co = draw.shape[1]
        if hip.image_conv_wgrad_ok(self.img, draw) and self.e.h1 <= 128:
            # one pass over draw and the image on the lane: no patch matrix (the bias column is not needed: dbeta comes from the BatchNorm backward)
            grads[wkey] = gradgemm.on_lane(
                lambda: hip.image_conv_wgrad(self.img, draw)[0].reshape(co, 3, 3, 3).permute(0, 3, 1, 2).contiguous(), draw)
        else:
            grads[wkey] = _wgrad_async(draw, self.bw._image_cols(), pix,
                                       post=lambda d: d[:, :27].reshape(co, 3, 3, 3).permute(0, 3, 1, 2).contiguous())

    def adapter(self, grads, j, dsum, x_pre):
        """-> the gradient matrix to hand to msclip_adapter_dx together with the RAW depthwise filter."""
        e, w, Bi = self.e, self.w, self.Bi
        a = e.adapters[j]
        g, D, C, k, hw, L = e.g, e.D, a["C"], a["k"], e.par_hw[j], e.Lv
        g2 = g * g
        p = f"visual.transformer.parallel_lateral_adapter.{j}"
        sv = self.saved[p + ".bottom_dw_conv.bn"]
        if len(sv) == 6:
            # (options.TRAIN.adapter_bn_views) the BatchNorm backward over whole samples as rows of L * D columns: the class token's
            # columns get mean 0 / rstd 1 / gamma 1 and no share of the sums, i.e. dx = dy there -- one pass writes the whole matrix
            # that msclip_adapter_dx reads; no gather of the grid rows, no clone, no scatter back
            x_full, mean, rstd, gam, n, _ = sv
            dfull = torch.empty_like(dsum)
            dg, db = hip.bn_bwd_token_columns(dsum.view(Bi, L * D), x_full.view(Bi, L * D), mean, rstd, gam, dfull.view(Bi, L * D), L, n)
            grads[p + ".bottom_dw_conv.bn.weight"], grads[p + ".bottom_dw_conv.bn.bias"] = dg, db
            dT_bf = hip.cast_bf16_colsum(dsum, fold=False, skip_group=g2)[0]
        else:
            dT = dsum.view(Bi, L, D)[:, 1:].reshape(Bi * g2, D)
            # bottom: BN(dw3x3(grid)) with batch statistics
            draw = self._bn_bwd(grads, p + ".bottom_dw_conv.bn", dT)
            dfull = dsum.clone()
            dfull.view(Bi, L, D)[:, 1:] = draw.view(Bi, g2, D)
            dT_bf = hip.cast_bf16(dT)
        grads[p + ".bottom_dw_conv.conv.weight"] = hip.dw3x3_wgrad(dfull, x_pre, Bi, L, g).t().reshape(D, 1, 3, 3)
        # top-down: T = Wp . BN(dwpool(par[j]))
        pw = self.raw.pw[j]
        grads[p + ".top2bottom_pw_conv.conv.weight"] = _wgrad_async(dT_bf, w["pool"][j], Bi * g2,
                                                                    post=lambda d: d.reshape(D, C, 1, 1))
        wt = pw.weight[:, :C].t().contiguous()
        dt = _zbuf(Bi * g2, C, e.dev)
        hip.gemm(dT_bf, wt, dt)
        dpool = self._bn_bwd(grads, p + ".top2bottom_dw_conv.bn", dt)
        grads[p + ".top2bottom_dw_conv.conv.weight"] = hip.dwpool_wgrad(dpool, w["par"][j], Bi, hw, hw, C, k).t().reshape(C, 1, k, k)
        if self.bw.dpar[0] is None:
            self.bw.dpar[0] = _zbuf(Bi * hw * hw, C, e.dev)
            hip.dwpool_bwd(dpool, self.raw.pool[j], self.bw.dpar[0], Bi, hw, hw, C, k)
        else:
            hip.dwpool_bwd(dpool, self.raw.pool[j], self.bw.dpar[0], Bi, hw, hw, C, k, accumulate=True)
        self._stage_bwd(grads, j)
        return dfull, self.raw.dww[j]

    def _stage_bwd(self, grads, j):
        e, w = self.e, self.w
        da, db_ = self.bw.dpar
        self.bw.dpar = [None, None]
        if j == 0:
            self._first(grads, "visual.transformer.parallel_branch_v.0.conv.weight", "visual.transformer.parallel_branch_v.0.bn", da,
                        y=w["par"][j], dy2=db_)
            return
        c1, c2, cr, c3, q = self.raw.par[j]
        t1, t2, _ = w["par_tmp"][j]
        src = w["par"][j - 1]
        d3, dr = self._bn_bwd_masked(grads, [q + ".bn3", q + ".residual_bn"], da, y=w["par"][j], dy2=db_)
        del da, db_
        dt2 = self._conv(grads, ("par", j, 3), c3, q + ".conv3.weight", t2, d3)
        short = self.bw._shortcut_ok(cr)     # the shortcut's input gradient is added into conv1's below (no map of its own)
        dsrc_a = self._conv(grads, ("par", j, "r"), cr, q + ".residual_conv.weight", src, dr, need_dx=not short)
        del d3
        if not short:
            del dr
        d2, = self._bn_bwd_masked(grads, [q + ".bn2"], dt2, y=t2)
        dt1 = self._conv(grads, ("par", j, 2), c2, q + ".conv2.weight", t1, d2, relu_of=t1)      # (through t1's ReLU)
        del d2, dt2
        d1 = self._bn_bwd(grads, q + ".bn1", dt1)
        dsrc_b = self._conv(grads, ("par", j, 1), c1, q + ".conv1.weight", src, d1)
        if short:
            self.bw._shortcut_dgrad_into(("par", j, "r"), cr, dr, dsrc_b, self.Bi)
            dsrc_a, dsrc_b = dsrc_b, None
        self.bw.dpar = [dsrc_a, dsrc_b]

    def stem(self, grads, dtok):
        e, w, Bi = self.e, self.w, self.Bi
        g2, D = e.g * e.g, e.D
        sp = "visual.transformer.resblocks.0"
        dlast = hip.cast_bf16_colsum(dtok[:Bi * e.Lv], fold=False, skip_group=g2)[0]     # the grid rows (class rows skipped), one pass, no gather copy
        x = w["stem"][-1]
        cin = x.shape[1]
        grads[sp + ".last_conv.weight"] = _wgrad_async(dlast, x[:Bi * g2], Bi * g2, post=lambda d: d.reshape(D, cin, 1, 1))
        dy = _zbuf(Bi * g2, x.shape[1], e.dev)
        hip.gemm(dlast, e.w_last.t().contiguous(), dy)
        dy2 = None
        for i in reversed(range(len(self.raw.stem))):
            main, short, q = self.raw.stem[i]
            x_in = w["stem"][i - 1] if i else w["S1"]
            dm, ds = self._bn_bwd_masked(grads, [q + ".bn1", q + ".downsample.1"], dy, y=w["stem"][i], dy2=dy2)
            dy = self._conv(grads, ("stem", i, "m"), main, q + ".conv1.weight", x_in, dm)
            if self.bw._shortcut_ok(short):              # the 1x1 / stride-2 shortcut's input gradient: added into the main path's map
                self._conv(grads, ("stem", i, "s"), short, q + ".downsample.0.weight", x_in, ds, need_dx=False)
                self.bw._shortcut_dgrad_into(("stem", i, "s"), short, ds, dy, Bi)
                dy2 = None
            else:
                dy2 = self._conv(grads, ("stem", i, "s"), short, q + ".downsample.0.weight", x_in, ds)
        self._first(grads, sp + ".conv1.weight", sp + ".bn1", dy, y=w["S1"], dy2=dy2)
        self.bw.col_img = None
        self.saved = {}                                  # the raw maps (several GB at batch 512) are not kept between steps
