"""Training step of the MS-CLIP-S hot path: forward that keeps what the backward needs, backward of every parameter,
AdamW.

The reference release contains no trainer and no loss; what is differentiated here is the forward it defines
(lib/models/clip_openai_pe_res_v1.py, "M.py") and the symmetric cross-entropy this build adds, with the reference's
gather semantics (gradient only through the local rows, lib/utils/comm.py:151-152) and its optimizer hyper-parameters
(separate LR / weight decay for the shared tensors: experiments/model/*-msclips.yaml `LR_SHARE` / `WD_SHARE`, scaled
with the world size in lib/config/default.py:299-304; no decay on bias / LayerNorm / BatchNorm, `WITHOUT_WD_LIST`).

This file: the token side -- the contrastive head (fused LSE sweeps forward; dL/dlogits blocks + GEMMs backward), L2
norm, both projections, ln_post / ln_final; all transformer blocks of both towers (LayerNorm, QKV / out_proj / c_fc /
c_proj dgrad + wgrad through msclip_gemm, msclip_attention_bwd, QuickGELU), where the modality-shared tensors
(M.py:2808-2830) receive the SUM of the image- and text-row gradients because the towers' tokens are rows of one matrix
and one wgrad GEMM contracts over all of them; the token path of the lateral adapters, ln_pre, class / positional /
token embeddings; the AdamW step and the rank averaging (comm.GradReducer).  train_conv.py: the convolutional side
(stem, parallel branch, adapter convolutions) with train-mode BatchNorm (bn="batch": per-GPU batch statistics, running
statistics updated) or frozen statistics (bn="frozen").  Together: all 325 gradient tensors of both released configs.

Parity: tests/test_gpu_train.py compares every gradient with autograd of the REAL reference
(tests/golden/*.grads.npz in eval() mode, b32-yfcc-msclips.grads_trainbn.npz in train() mode; captured by
tools/make_golden.py::grads_fixture).
"""

import torch
import torch.distributed as dist

from . import comm as C
from . import hip
from . import gradgemm
from . import options
from .gradgemm import dgrad as _dgrad, wgrad as _wgrad, wgrad_async as _wgrad_async
from .train_conv import ConvSideBackward, ConvSideBatchNorm

BF = torch.bfloat16
F32 = torch.float32
_LEAVES = ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
           "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")


class TrainStep:
    """forward + backward (+ AdamW step) for one local batch.  `engine` is the model's msclip_amd.engine.Engine."""

    def __init__(self, model, lr=None, lr_share=None, wd=0.05, wd_share=None, betas=(0.9, 0.999), eps=1e-8, bn="frozen",
                 without_wd=("bn", "bias", "ln")):
        """bn = "frozen": BatchNorm with its running statistics (gamma / beta trained; the inference kernels' folded
        form); bn = "batch": train-mode BatchNorm -- per-GPU batch statistics in the forward, their backward, running
        statistics updated with momentum 0.1 (what the reference's modules do in train()).
        Optimizer defaults are the reference yaml's (experiments/model/b32.yaml:39-50: adamW, WD 0.05, no decay on
        'bn' / 'bias' / 'ln'; no OPTIMIZER_ARGS => torch.optim.AdamW's betas (0.9, 0.999) and eps 1e-8).
        `without_wd`: TRAIN.WITHOUT_WD_LIST keywords ('bn': BatchNorm parameters, 'ln': LayerNorm parameters, 'bias': names
        ending in 'bias')."""
        assert bn in ("frozen", "batch")
        unknown = set(without_wd) - {"bn", "bias", "ln", "gn", "dw"}
        if unknown:
            raise NotImplementedError(f"TRAIN.WITHOUT_WD_LIST keywords {sorted(unknown)} are not implemented")
        self.without_wd = tuple(without_wd)
        self.bn = bn
        self.convbn = None
        self.model = model
        self.eng = model.engine()
        self.dev = self.eng.dev
        self.lr, self.lr_share = lr, lr_share
        self.wd, self.wd_share = wd, wd_share
        self.betas, self.eps = betas, eps
        self.state = {}
        self.steps = 0

    # ------------------------------------------------------------------ forward that keeps what the backward needs
    @hip.off_default_stream
    def forward(self, img, tok):
        try:
            return self._forward(img, tok)
        finally:
            self.eng.force_unfused = False

    def _forward(self, img, tok):
        e = self.eng
        e.refresh()
        with torch.cuda.device(e.dev), torch.no_grad():
            Bi, Bt = img.shape[0], tok.shape[0]
            assert Bi == Bt, "a training step needs image-text pairs"
            if e.Lv > 208 or e.Lt > 208:
                raise NotImplementedError("msclip_attention_bwd covers sequences up to 208 tokens")
            prev = getattr(self, "saved", None)
            if prev is not None:                                   # a forward that was never differentiated: release its maps
                prev["w"].pop("held", None)
                self.saved = None
            w = e._workspace(Bi, Bt)
            # the conv side's maps (w["stem"], w["par"], w["pool"], S1, P0) are read again by backward(): until then the
            # engine's inference entry points (an eval / logging call of the same shape in between) get another workspace
            w["held"] = True
            # captions run packed like the inference path (engine.text_pack_enabled): rows behind a caption's EOT position
            # neither influence an output nor receive a gradient.  `tok` may be a batch staged ahead (engine.stage_captions).
            from .engine import Captions
            pack = e.text_pack_enabled()
            cap = tok if isinstance(tok, Captions) else None
            tokc = cap.tok if cap is not None else e._check_tok(tok)
            if pack and cap is None:
                cap = e.stage_captions(tokc)
            if not pack:
                cap = None
                e._text_unpacked(w, Bt)
            else:
                e._text_sizes(cap, w, Bt)                   # the host read (immediate for a batch staged a step ahead)
            Mv = w["Mv"]
            D = e.D
            X = w["X"]
            sv = dict(Bi=Bi, Bt=Bt, Mv=Mv, layers=[None] * e.n_layers, tok=tokc, w=w, cap=cap)
            # ---- fronts (the conv side's maps stay in the workspace `w`; the tokens in front of ln_pre are cloned)
            keep = []
            sv["img"] = e._check_img(img)
            e.force_unfused = True                   # layer-by-layer conv side: every map the backward reads stays in `w`
            cb = None
            if self.bn == "batch":
                if self.convbn is None:
                    self.convbn = ConvSideBatchNorm(self)
                cb = self.convbn
                cb.begin(sv["img"], w, Bi)
                cb.front()
            e._vision_front(sv["img"], w, Bi, keep_pre=keep, convs_done=cb is not None)
            sv["tok_pre"] = keep[0]
            g2 = e.g * e.g
            # frozen statistics: the image-only conv branch + the adapters' top-down halves on the engine's side stream, as in
            # the inference schedule (layer by layer: force_unfused keeps every map the backward reads in the workspace)
            conv_events = None
            # (same-box A/B: 89.6-90.2 -> 88.5-88.9 ms per step)
            if (cb is None and not gradgemm._ranks_share_a_gpu() and not options.TRAIN.wgrad_sync
                    and e.lateral == sorted(e.lateral)):
                conv_events = e._conv_branch_on_side_stream(w, Bi)
            if (cb is not None and conv_events is None and not gradgemm._ranks_share_a_gpu() and not options.TRAIN.wgrad_sync
                    and e.lateral == sorted(e.lateral)):
                # train-mode BatchNorm: the same for the raw-conv -> statistics -> normalise chains of the parallel branch and
                # the adapters' top-down halves (they depend on the image only)
                if w["Ts"] is None:
                    w["Ts"] = [torch.empty(Bi * g2, D, dtype=F32, device=e.dev) for _ in e.adapters]
                cur, side = torch.cuda.current_stream(e.dev), C.side_stream(e.dev)
                ready = torch.cuda.Event()
                ready.record(cur)
                side.wait_event(ready)
                conv_events = []
                with torch.cuda.stream(side):
                    for j in range(len(e.adapters)):
                        cb.stage(j)
                        cb.adapter_top(j, w["Ts"][j])
                        ev = torch.cuda.Event()
                        ev.record(side)
                        conv_events.append(ev)
            M = sv["M"] = w["M"]
            sv.update(Lmax=w["Lmax"], pad=w["pad"], Mt_live=w["Mt_live"])
            e._text_front(sv["tok"], w, Bt)
            # ---- blocks.  Xc = the residual matrix the next layer reads: the workspace's X at first; every layer that runs over
            # all rows writes its two residual updates into fresh matrices (the backward needs the layer's input and its
            # post-attention state: they stay where they are instead of being cloned), so Xc moves on.
            Xc = X
            for i in range(e.n_layers):
                vb = e.vblk[i]
                tb = e.tblk[i]
                L = dict(adapter=None)
                if vb is not None and i in e.lateral:
                    j = e.lateral.index(i)
                    a = e.adapters[j]
                    asum = torch.empty(Mv, D, dtype=F32, device=e.dev)
                    if cb is not None and conv_events is not None:
                        torch.cuda.current_stream(e.dev).wait_event(conv_events[j])
                        cb.adapter_sum(j, Xc[:Mv], w["Ts"][j], asum)
                    elif cb is not None:
                        cb.stage(j)
                        cb.adapter_top(j, w["T"])
                        cb.adapter_sum(j, Xc[:Mv], w["T"], asum)
                    elif conv_events is not None:
                        torch.cuda.current_stream(e.dev).wait_event(conv_events[j])
                        hip.adapter_sum(Xc[:Mv], w["Ts"][j], a["dww"], a["dwb"], asum, Bi, e.Lv, e.g, e.usecls)
                    else:
                        e._parallel_stage(j, w, Bi)
                        hip.dwpool(w["par"][j], a["pool"], w["pool"][j], Bi, e.par_hw[j], e.par_hw[j], a["C"], a["k"])
                        hip.gemm(w["pool"][j], a["pw"].weight, w["T"], M=Bi * g2, N=a["pw"].cout, bias=a["pw"].bias, ldx=a["pw"].cin)
                        hip.adapter_sum(Xc[:Mv], w["T"], a["dww"], a["dwb"], asum, Bi, e.Lv, e.g, e.usecls)
                    x_pre = Xc[:Mv].clone()                                                   # what the depthwise 3x3 read
                    hip.layernorm(asum, a["ln"].g, a["ln"].b, Xc[:Mv], Mv)                    # X[:Mv] <- ln_adapt(sum), fp32
                    L["adapter"] = dict(j=j, sum=asum, x_pre=x_pre)
                segs = ([(0, Mv, vb)] if vb is not None else []) + [(Mv, M, tb)]
                r_lo = segs[0][0]
                fresh = r_lo == 0                         # (text block 0 leaves the image rows alone: in place, cloned)
                L["x_in"] = Xc if fresh else Xc[r_lo:M].clone()
                # Round 6: behind the LAST block only x[:, 0] of every image (M.py:2685) and the EOT row of every caption (M.py:3057-3060)
                # are read, and out_proj / ln_2 / c_fc / c_proj are row-wise: they run -- forward here, backward in backward() -- on
                # those Bi + Bt rows instead of on every token (what the inference path has done since round 3; the rows' gradients
                # are the only non-zero ones behind the block, so every parameter gradient is unchanged)
                compact = (i == e.n_layers - 1 and len(segs) == 2 and L["adapter"] is None and options.TRAIN.compact_last_block)
                XM = None if compact else (torch.empty(M, D, dtype=F32, device=e.dev) if fresh else Xc)
                lno1 = torch.empty(M, D, dtype=BF, device=e.dev)
                if len(segs) == 2:                          # both towers' rows in one launch (modality-specific gamma / beta per row segment)
                    hip.layernorm_split(Xc[:M], vb["ln1"].g, vb["ln1"].b, tb["ln1"].g, tb["ln1"].b, Mv, lno1[:M], M)
                else:
                    for r0, r1, b in segs:
                        hip.layernorm(Xc[r0:r1], b["ln1"].g, b["ln1"].b, lno1[r0:r1], r1 - r0)
                groups = [(r_lo, M, segs[0][2]["w"])] if len(segs) == 2 and vb["w"] is tb["w"] else \
                         [(r0, r1, b["w"]) for r0, r1, b in segs]
                qkv = torch.empty(M, 3 * D, dtype=BF, device=e.dev)
                ao = torch.empty(M, D, dtype=BF, device=e.dev)
                for r0, r1, bw in groups:
                    hip.gemm(lno1[r0:r1], bw.wqkv, qkv[r0:r1], bias=bw.bqkv)
                if vb is not None:
                    hip.attention(qkv[:Mv], ao[:Mv], Bi, e.Lv, e.heads, False)
                e._attention_text(w, qkv, ao, Bt)
                if compact:
                    nc = Bi + Bt
                    crow = torch.cat([torch.arange(0, Mv, e.Lv, dtype=torch.int32, device=e.dev), w["eot"][:Bt]])   # live rows of X
                    xin_c, ao_c = torch.empty(nc, D, dtype=F32, device=e.dev), torch.empty(nc, D, dtype=BF, device=e.dev)
                    hip.gather_rows(Xc, xin_c, nc, row_idx=crow)
                    hip.gather_rows(ao, ao_c, nc, row_idx=crow)
                    cgroups = [(0, nc, vb["w"])] if vb["w"] is tb["w"] else [(0, Bi, vb["w"]), (Bi, nc, tb["w"])]
                    xm_c = torch.empty(nc, D, dtype=F32, device=e.dev)
                    for r0, r1, bw in cgroups:
                        hip.gemm(ao_c[r0:r1], bw.wo, xm_c[r0:r1], bias=bw.bo, resid=xin_c[r0:r1], resid_kind=hip.RESID_F32)
                    lno2_c = torch.empty(nc, D, dtype=BF, device=e.dev)
                    hip.layernorm_split(xm_c, vb["ln2"].g, vb["ln2"].b, tb["ln2"].g, tb["ln2"].b, Bi, lno2_c, nc)
                    h_c, hid_c = torch.empty(nc, 4 * D, dtype=BF, device=e.dev), torch.empty(nc, 4 * D, dtype=BF, device=e.dev)
                    xn_c = torch.empty(nc, D, dtype=F32, device=e.dev)
                    for r0, r1, bw in cgroups:
                        hip.gemm(lno2_c[r0:r1], bw.wfc, h_c[r0:r1], bias=bw.bfc)
                        hip.quickgelu(h_c[r0:r1], hid_c[r0:r1])
                        hip.gemm(hid_c[r0:r1], bw.wpr, xn_c[r0:r1], bias=bw.bpr, resid=xm_c[r0:r1], resid_kind=hip.RESID_F32)
                    L.update(r_lo=r_lo, segs=segs, groups=groups, lno1=lno1, qkv=qkv, ao=ao,
                             compact=dict(crow=crow, cgroups=cgroups, ao=ao_c, x_mid=xm_c, lno2=lno2_c, h=h_c, hid=hid_c, x_out=xn_c))
                    sv["layers"][i] = L
                    sv["compact"] = xn_c
                    continue
                for r0, r1, bw in groups:
                    hip.gemm(ao[r0:r1], bw.wo, XM[r0:r1], bias=bw.bo, resid=Xc[r0:r1], resid_kind=hip.RESID_F32)
                L["x_mid"] = XM if fresh else XM[r_lo:M].clone()
                XN = torch.empty(M, D, dtype=F32, device=e.dev) if fresh else Xc
                lno2 = torch.empty(M, D, dtype=BF, device=e.dev)
                if len(segs) == 2:
                    hip.layernorm_split(XM[:M], vb["ln2"].g, vb["ln2"].b, tb["ln2"].g, tb["ln2"].b, Mv, lno2[:M], M)
                else:
                    for r0, r1, b in segs:
                        hip.layernorm(XM[r0:r1], b["ln2"].g, b["ln2"].b, lno2[r0:r1], r1 - r0)
                h = torch.empty(M, 4 * D, dtype=BF, device=e.dev)
                hid = torch.empty(M, 4 * D, dtype=BF, device=e.dev)
                for r0, r1, bw in groups:
                    if (r1 - r0) % 256 == 0:
                        # ONE launch writes the pre-activation (the backward's QuickGELU' needs it) and the activation
                        # (the ping-pong kernel's training forms cover whole 256-row tiles: any other row count takes two launches)
                        hip.gemm(lno2[r0:r1], bw.wfc, hid[r0:r1], bias=bw.bfc, act=hip.ACT_QUICKGELU, out2=h[r0:r1])
                    else:
                        hip.gemm(lno2[r0:r1], bw.wfc, h[r0:r1], bias=bw.bfc)
                        hip.quickgelu(h[r0:r1], hid[r0:r1])
                for r0, r1, bw in groups:
                    hip.gemm(hid[r0:r1], bw.wpr, XN[r0:r1], bias=bw.bpr, resid=XM[r0:r1], resid_kind=hip.RESID_F32)
                L.update(r_lo=r_lo, segs=segs, groups=groups, lno1=lno1, qkv=qkv, ao=ao, lno2=lno2, h=h, hid=hid)
                sv["layers"][i] = L
                Xc = XN
            e.force_unfused = False
            if cb is not None:
                cb.update_running_stats()
            # ---- heads + loss
            if sv.get("compact") is not None:
                # the heads read the compact matrix of the last block's live rows (image cls rows, then EOT rows)
                sv["x_out"] = sv["compact"]
                xc_own, w["XC"] = w["XC"], sv["compact"]
                try:
                    e._head_image(w, Bi, compact=True)
                    e._head_text(w, Bt, compact=True, Bi=Bi)
                finally:
                    w["XC"] = xc_own
            else:
                sv["x_out"] = Xc if Xc is not X else X[:M].clone()
                w["X"] = Xc                              # the heads read the final residual matrix
                try:
                    e._head_image(w, Bi)
                    e._head_text(w, Bt)
                finally:
                    w["X"] = X
            sv.update(hv=w["hv"].clone(), ht=w["ht"].clone(), fv_raw=w["fv_raw"].clone(), ft_raw=w["ft_raw"].clone(),
                      fv=w["fv"].clone(), ft=w["ft"].clone(), fvb=w["fvb"].clone(), ftb=w["ftb"].clone(), eot=w["eot"].clone())
            # ---- loss.  The inference path forms its logits from bf16 unit features (error ~0.03 on a logit at T = 1/0.07:
            # fine for a loss value, but it perturbs every softmax probability by a few percent and with it the whole
            # gradient).  The training step splits each fp32 feature into bf16 hi + lo parts and contracts
            # [hi | hi | lo] . [hi | lo | hi] in one K = 3E GEMM: logits to ~2^-16 relative, still on the bf16 MFMA path.
            def split(f):
                hi = hip.cast_bf16(f)
                lo = hip.cast_bf16(f - hi.float())
                return torch.cat([hi, lo], dim=1)                                              # [B, 2E]: the gather payload
            # exp(logit_scale) stays ON THE DEVICE here: as a host scalar (the GEMM's alpha) it would have to be read back after
            # the previous step's optimizer update, i.e. the host would wait at this point of every forward until the GPU
            # has finished the step before (measured: forward returned after 83 ms instead of 5 with a step queued) and
            # then enqueue the backward against a GPU that is already running.  The image features carry the scale instead.
            s_dev = e.model.logit_scale.detach().float().exp()
            sv["scale"] = s_dev
            pi, pt = split(sv["fv"] * s_dev), split(sv["ft"])
            if C.comm.collectives:
                allpi, hi_ = C.gather_rows_async(pi)
                allpt, ht_ = C.gather_rows_async(pt)
                hi_.wait(); ht_.wait()
            else:
                allpi, allpt = pi, pt
            E = e.E
            n = allpi.shape[0]
            off = C.local_label_offset(Bi) if n > Bi else 0
            sv["off"], sv["n"] = off, n
            sv["allI"], sv["allT"] = allpi[:, :E].contiguous(), allpt[:, :E].contiguous()     # hi parts: dgrad operands (allI scaled)

            def logits_block(a, ball):                                                         # A_loc @ B_all^T, fp32 (scale in the image side)
                a3 = torch.cat([a[:, :E], a[:, :E], a[:, E:]], dim=1)
                b3 = torch.cat([ball[:, :E], ball[:, E:], ball[:, :E]], dim=1)
                S = torch.empty(a.shape[0], n, dtype=F32, device=e.dev)
                hip.gemm(a3, b3, S)
                return S
            S_i, S_t = logits_block(pi, allpt), logits_block(pt, allpi)                        # image rows / caption rows
            lse = torch.empty(2, Bi, dtype=F32, device=e.dev)
            hip.lse_rows(S_i, lse[0])
            hip.lse_rows(S_t, lse[1])
            out = torch.empty(1, dtype=F32, device=e.dev)
            hip.clip_loss_partial(lse[0], lse[1], S_i, off, 1.0 / (2.0 * n), out)
            sv["coll"] = n > Bi or C.comm.collectives
            if sv["coll"]:
                dist.all_reduce(out)
                lse_flat = torch.empty((n // Bi) * 2, Bi, dtype=F32, device=e.dev)            # concatenated along dim 0: every backend's form
                dist.all_gather_into_tensor(lse_flat, lse.contiguous())
                lse_all = lse_flat.view(n // Bi, 2, Bi).permute(1, 0, 2).reshape(2, n).contiguous()   # rank-major global order
            else:
                lse_all = lse
            sv.update(S_i=S_i, S_t=S_t, lse_loc=lse, lse_all=lse_all)
            self.saved = sv
            return out[0].clone()

    # ------------------------------------------------------------------ backward
    @hip.off_default_stream
    def backward(self, reduce=True, bucket_bytes=64 << 20, clone=False):
        """-> {reference state_dict key: fp32 gradient} for every parameter of the slice (shared tensors under their
        visual.* key; the text-tower aliases are the same Parameter objects).  Under N > 1 ranks (and reduce=True) the
        gradients are averaged over the ranks as the reference's DDP wrapper would: every gradient goes into a
        comm.GradReducer bucket the moment it exists, and the buckets' RCCL all-reduces run on the side stream under
        the rest of the backward (last block's bucket first).
        LIFETIME: at world size > 1 the returned tensors are views into the per-device gradient-bucket arena, which the NEXT
        backward() on this device overwrites (in-place 1 / world scaling and split-K writes included): call step() before the
        next backward, or pass clone=True to own the tensors (gradient accumulation over micro-batches, comparing two backward
        passes, two TrainSteps on one device).  step() refuses stale views (comm.GradViews.check_fresh).  At world size 1 the
        gradients are fresh tensors either way."""
        e, sv = self.eng, getattr(self, "saved", None)
        if sv is None:
            raise RuntimeError("TrainStep.backward() needs the activations of a TrainStep.forward() that has not been "
                               "differentiated yet (call forward first; one backward per forward)")
        dev, D, E = e.dev, e.D, e.E
        with torch.cuda.device(dev), torch.no_grad():
            Bi, Bt, Mv, M, n, off = sv["Bi"], sv["Bt"], sv["Mv"], sv["M"], sv["n"], sv["off"]
            reducer = C.GradReducer(bucket_bytes) if (reduce and C.comm.collectives) else None
            self.reducer = reducer

            class _Grads(dict):
                def __setitem__(self, k, v):
                    dict.__setitem__(self, k, v)
                    if reducer is not None:
                        reducer.add(k, v)

                def slot(self, k, shape):
                    """Where gradient k should be written if it can be born inside its all-reduce bucket (else None)."""
                    return reducer.reserve(k, shape, dev) if reducer is not None else None
            grads = _Grads()
            # Column sums nothing on the critical path reads (LayerNorm parameter gradients, bias gradients): their producers leave
            # per-block partial matrices, ONE msclip_colsum_multi launch folds all of them behind the transformer's backward (round 5:
            # ~120 msclip_colsum launches per step, 1.1 ms of the main queue)
            folds = hip.FoldPlan(dev)

            def fold_into(part, then, scale_n=0, scale=1.0):
                folds.add(part, then, scale_n=scale_n, scale=scale)

            def ln_param_grads(part, key):
                """part: (dgamma | dbeta) partials [blocks, 2 C] of a LayerNorm backward -> grads[key.weight], grads[key.bias]."""
                def then(r):
                    c = r.shape[0] // 2
                    grads[key + ".weight"], grads[key + ".bias"] = r[:c], r[c:]
                fold_into(part, then)

            def set_bias(key, val):
                """val: the bias gradient itself, or a 2-D matrix of partial sums still to be folded."""
                if val.dim() == 2:
                    fold_into(val, lambda r: grads.__setitem__(key, r))
                else:
                    grads[key] = val
            # the four wide weight gradients of a block: operand transposes on the lane at once, the split-K GEMM of job k
            # on THIS stream when job k + 1 is created (its transposes have run under the kernels issued in between)
            pending = []

            def flush_wgrads():
                while pending:
                    k, job, shape = pending.pop(0)
                    grads[k] = job.finish(grads.slot(k, shape))

            def wide_wgrad(k, dy, x, rows, shape, post=None):
                flush_wgrads()
                pending.append((k, gradgemm.WgradJob(dy, x, rows, post=post), shape))
            # ---- contrastive head: dL/dS blocks of this rank's image rows and caption rows
            npad = (n + 63) // 64 * 64
            wgt = 1.0 / (2.0 * n)

            def side(S, b_all, lse_row, lse_col, want_dscale):
                G = torch.empty(S.shape[0], npad, dtype=BF, device=dev)
                dsp = torch.empty(S.shape[0], dtype=F32, device=dev) if want_dscale else None
                hip.clip_loss_bwd_g(S, lse_row, lse_col, off, wgt, G, dsp)
                bt = hip.transpose_bf16(b_all, n, npad)                                        # [E, npad]
                d = torch.empty(S.shape[0], E, dtype=F32, device=dev)
                hip.gemm(G, bt, d)                                                             # dA = G @ B_all (scale: below)
                return d, dsp
            lse_i_loc, lse_t_loc = sv["lse_loc"][0], sv["lse_loc"][1]
            lse_i_all, lse_t_all = sv["lse_all"][0], sv["lse_all"][1]
            dfi, dsp = side(sv["S_i"], sv["allT"], lse_i_loc, lse_t_all, True)
            dfi *= sv["scale"]                                                                  # d(image features) = scale * G_i @ T_all
            dft, _ = side(sv["S_t"], sv["allI"], lse_t_loc, lse_i_all, False)                  # allI already carries the scale
            dscale = hip.colsum(dsp.view(-1, 1))                                               # sum_r sum_j G S
            if sv["coll"]:
                dist.all_reduce(dscale)
            # S already carries the scale: dL/dscale = sum G S / scale; logit_scale = log(scale) => dL/dlogit_scale = sum G S
            grads["logit_scale"] = dscale.reshape(())

            dX = torch.zeros(M, D, dtype=F32, device=dev)
            if self.bn == "batch":
                conv = self.convbn                       # holds the raw conv outputs / batch statistics of this forward
            else:
                conv = ConvSideBackward(self)
                conv.begin(sv["img"], sv["w"], Bi)

            compact_x = sv.get("compact")                 # [Bi + Bt, D]: the last block ran its row-wise tail on the live rows only
            dXC = torch.zeros(Bi + Bt, D, dtype=F32, device=dev) if compact_x is not None else None

            def head(feat_raw, dfeat, hrow, w_proj, ln, key_proj, key_ln, row_idx=None, row_mul=1, c0=None):
                dfr = torch.empty_like(feat_raw)
                hip.l2norm_bwd(feat_raw, dfeat, dfr)
                dfr_b = hip.cast_bf16(dfr)
                grads[key_proj] = _wgrad(hrow, dfr_b, hrow.shape[0])                           # [D, E] like the parameter
                dh = torch.empty(hrow.shape[0], D, dtype=F32, device=dev)                      # fp32: it feeds column sums
                hip.gemm(dfr_b, w_proj.t().contiguous(), dh)                                   # dfr [B, E] @ W [E, D]
                if c0 is not None:                        # compact rows c0 .. of x_out / dXC, one per sample
                    n_ = hrow.shape[0]
                    part, _ = hip.layernorm_bwd(sv["x_out"][c0:c0 + n_], dh, ln.g, dXC[c0:c0 + n_], n_, fold=False)
                else:
                    part, _ = hip.layernorm_bwd(sv["x_out"], dh, ln.g, dX, hrow.shape[0], row_idx=row_idx, row_mul=row_mul, fold=False)
                ln_param_grads(part, key_ln)
            if compact_x is not None:
                head(sv["fv_raw"], dfi, sv["hv"], e.w_vproj, e.ln_post, "visual.proj", "visual.ln_post", c0=0)
                head(sv["ft_raw"], dft, sv["ht"], e.w_tproj, e.ln_final, "text_projection", "ln_final", c0=Bi)
            else:
                head(sv["fv_raw"], dfi, sv["hv"], e.w_vproj, e.ln_post, "visual.proj", "visual.ln_post", row_mul=e.Lv)
                head(sv["ft_raw"], dft, sv["ht"], e.w_tproj, e.ln_final, "text_projection", "ln_final", row_idx=sv["eot"])

            def cast_with_bias_sums(dX, dY, r_lo, groups):
                """dY[r_lo:M] = bf16(dX[r_lo:M]) -- the operand of a projection's dgrad / wgrad GEMMs -- and {id(block
                weights): column sums of the group's rows} = that projection's bias gradient; one pass over dX when a
                single (shared) weight set covers all rows."""
                if len(groups) == 1 and groups[0][0] == r_lo and groups[0][1] == M:
                    _, s = hip.cast_bf16_colsum(dX[r_lo:M], dY[r_lo:M], fold=False)    # partial sums: folded with the others (set_bias)
                    return {id(groups[0][2]): s}
                hip.cast_bf16(dX[r_lo:M], dY[r_lo:M])
                return {id(bw): hip.colsum(dX[r0:r1]) for r0, r1, bw in groups}

            # ---- the dgrad GEMMs' W operands: W^T of every projection (bf16 [K, N]; the weights changed in the last optimizer
            # step), all of them up front on the lane stream with this library's transpose kernel, last block first, one event
            # per block -- not four ATen transposed copies per block on the main queue (74 launches, 1.8 ms per step)
            wts, wt_ready = {}, {}
            if not (options.TRAIN.wgrad_sync or gradgemm._ranks_share_a_gpu()):
                cur_s, ln_s = torch.cuda.current_stream(dev), gradgemm.lane(dev)
                ev0 = torch.cuda.Event()
                ev0.record(cur_s)
                ln_s.wait_event(ev0)
                sets = []
                for i in reversed(range(e.n_layers)):
                    for blk in (e.tblk[i], e.vblk[i]):
                        if blk is not None and all(blk["w"] is not b for b in sets):
                            sets.append(blk["w"])
                srcs = [t for bw in sets for t in (bw.wpr, bw.wfc, bw.wo, bw.wqkv)]
                multi = all(t.shape[0] % 64 == 0 for t in srcs)
                with torch.cuda.stream(ln_s):
                    if multi:
                        # ONE table-driven launch for all of them (48 launches one by one kept the host busy for 0.6 ms at the start
                        # of the backward, the main queue idle behind it); the table lives while the packed weights do not move
                        plan = getattr(self, "_wt_plan", None)
                        if plan is None or plan.key != tuple(t.data_ptr() for t in srcs):
                            plan = self._wt_plan = hip.TransposePlan(srcs)
                        outs = plan.run()
                        ready = torch.cuda.Event()
                        ready.record(ln_s)
                        for j, bw in enumerate(sets):
                            wts[id(bw)] = tuple(outs[4 * j:4 * j + 4])
                            wt_ready[id(bw)] = ready             # (one launch, one event: whichever weight set is asked for first waits for it)
                    else:
                        for bw in sets:
                            wts[id(bw)] = tuple(hip.transpose_bf16(t, t.shape[0], t.shape[0]) for t in (bw.wpr, bw.wfc, bw.wo, bw.wqkv))
                            for t in wts[id(bw)]:
                                t.record_stream(cur_s)
                            wt_ready[id(bw)] = torch.cuda.Event()
                            wt_ready[id(bw)].record(ln_s)

            def w_t(bw, which):
                """W^T of block weights bw: 0 c_proj [4D, D]^T.., 1 c_fc, 2 out_proj, 3 in_proj (packed: q rows scaled)."""
                if id(bw) in wts:
                    ev = wt_ready.pop(id(bw), None)
                    if ev is not None:
                        torch.cuda.current_stream(dev).wait_event(ev)
                    return wts[id(bw)][which]
                return (bw.wpr, bw.wfc, bw.wo, bw.wqkv)[which].t().contiguous()

            def ln_bwd_segments(x_saved, dlno, segs, groups, r_lo, which, i, dY_next):
                """LayerNorm backward (`which` = "ln1" / "ln2" of block i) of every row segment: dX += its input gradient.  With
                dY_next (bf16 [M, D]) the same pass leaves bf16(new dX) there -- the output gradient of the projection in front of
                this LayerNorm point, operand of its dgrad / wgrad GEMMs -- and returns {id(block weights): that projection's bias
                gradient} from per-block column sums (no cast_bf16_colsum pass over dX); else returns None."""
                parts = {}
                for r0, r1, b in segs:
                    kw = {}
                    if dY_next is not None:
                        gid = next(id(bw) for g0, g1, bw in groups if g0 <= r0 and r1 <= g1)
                        first = gid not in parts
                        if first:
                            parts[gid] = torch.empty(hip.LN_PART_BLOCKS, D, dtype=F32, device=dev)
                        kw = dict(dxb=dY_next[r0:r1], sum_part=parts[gid], sum_accumulate=not first)
                    pre = f"visual.transformer.resblocks.{i}" if b is e.vblk[i] else f"transformer.resblocks.{i}"
                    part, _ = hip.layernorm_bwd(x_saved[r0 - r_lo:r1 - r_lo], dlno[r0:r1], b[which].g, dX[r0:r1], r1 - r0, fold=False, **kw)
                    ln_param_grads(part, f"{pre}.{'ln_1' if which == 'ln1' else 'ln_2'}")
                return parts if dY_next is not None else None          # (partial sums: set_bias folds them)

            # ---- blocks, last to first
            fuse_cast = True
            carry = None                      # (dY, bias sums) of this block's MLP half, left by the block above's ln_1 backward
            for i in reversed(range(e.n_layers)):
                L = sv["layers"][i]
                r_lo, segs, groups = L["r_lo"], L["segs"], L["groups"]
                names = {id(e.tblk[i]["w"]): f"transformer.resblocks.{i}"}
                if e.vblk[i] is not None:                    # shared tensors live under their visual.* name (one Parameter)
                    names[id(e.vblk[i]["w"])] = f"visual.transformer.resblocks.{i}"
                cm = L.get("compact")
                if cm is not None:
                    # the last block's row-wise tail ran on the Bi + Bt live rows (forward above): its backward on the same rows.  dXC holds
                    # the heads' gradient wrt the block's output at those rows; every other row's is zero.
                    assert carry is None
                    nc = Bi + Bt
                    xm_c, lno2_c, h_c, hid_c, ao_c = cm["x_mid"], cm["lno2"], cm["h"], cm["hid"], cm["ao"]
                    dlno2_c = torch.empty(nc, D, dtype=F32, device=dev)
                    dy_c = hip.cast_bf16(dXC)
                    for r0, r1, bw in cm["cgroups"]:
                        p = names[id(bw)]
                        grads[p + ".mlp.c_proj.weight"] = _wgrad(dy_c[r0:r1], hid_c[r0:r1], r1 - r0)
                        grads[p + ".mlp.c_proj.bias"] = hip.colsum(dXC[r0:r1])
                        dh_c = torch.empty(r1 - r0, 4 * D, dtype=BF, device=dev)
                        hip.quickgelu_bwd(h_c[r0:r1], _dgrad(dy_c[r0:r1], w_t(bw, 0)), dh_c)
                        grads[p + ".mlp.c_fc.weight"] = _wgrad(dh_c, lno2_c[r0:r1], r1 - r0)
                        grads[p + ".mlp.c_fc.bias"] = hip.colsum(dh_c)
                        _dgrad(dh_c, w_t(bw, 1), dlno2_c[r0:r1])
                    for (c0, c1), b in (((0, Bi), e.vblk[i]), ((Bi, nc), e.tblk[i])):
                        pre = f"visual.transformer.resblocks.{i}" if b is e.vblk[i] else f"transformer.resblocks.{i}"
                        part, _ = hip.layernorm_bwd(xm_c[c0:c1], dlno2_c[c0:c1], b["ln2"].g, dXC[c0:c1], c1 - c0, fold=False)
                        ln_param_grads(part, pre + ".ln_2")
                    # dXC is now the gradient wrt the rows behind the attention (x_mid): out_proj on the compact rows, then both
                    # results go back to their rows of the token matrix -- the attention output's gradient (zero elsewhere) and,
                    # through the residual connection, the block input's
                    dy2_c = hip.cast_bf16(dXC)
                    dao_c = torch.empty(nc, D, dtype=BF, device=dev)
                    for r0, r1, bw in cm["cgroups"]:
                        p = names[id(bw)]
                        grads[p + ".attn.out_proj.weight"] = _wgrad(dy2_c[r0:r1], ao_c[r0:r1], r1 - r0)
                        grads[p + ".attn.out_proj.bias"] = hip.colsum(dXC[r0:r1])
                        _dgrad(dy2_c[r0:r1], w_t(bw, 2), dao_c[r0:r1])
                    rows_c = cm["crow"].long()
                    dao = torch.zeros(M, D, dtype=BF, device=dev)
                    dao.index_copy_(0, rows_c, dao_c)
                    dX.index_copy_(0, rows_c, dXC)                   # (dX is still all zero here: nothing but the heads wrote a gradient yet)
                    dlno = torch.empty(M, D, dtype=F32, device=dev)
                    dqkv = torch.empty(M, 3 * D, dtype=BF, device=dev)
                else:
                    hid = L["hid"]
                    if carry is not None:
                        dY, bsum = carry
                        carry = None
                    else:
                        dY = torch.empty(M, D, dtype=BF, device=dev)
                        bsum = cast_with_bias_sums(dX, dY, r_lo, groups)
                    # gradients of the LayerNorm outputs stay fp32: they are only read by the LayerNorm backward, whose dbeta /
                    # dgamma are column sums of nearly cancelling terms (a bf16 dy costs 10-30 % on those sums at small batch)
                    dlno = torch.empty(M, D, dtype=F32, device=dev)
                    for r0, r1, bw in groups:
                        p = names[id(bw)]
                        wide_wgrad(p + ".mlp.c_proj.weight", dY[r0:r1], hid[r0:r1], r1 - r0, (D, 4 * D))
                        set_bias(p + ".mlp.c_proj.bias", bsum[id(bw)])
                    dh = torch.empty(M, 4 * D, dtype=BF, device=dev)
                    dh_part = {}
                    for r0, r1, bw in groups:
                        if (r1 - r0) % 256 == 0:
                            # dh = (dY . W_proj) * QuickGELU'(h): the activation's derivative in the dgrad GEMM's epilogue, which
                            # also leaves dh's column sums per 128 rows (c_fc's bias gradient without a second pass over dh)
                            dh_part[id(bw)] = torch.empty((r1 - r0) // 128, 4 * D, dtype=F32, device=dev)
                            hip.gemm(dY[r0:r1], w_t(bw, 0), dh[r0:r1], resid=L["h"][r0:r1], resid_kind=hip.RESID_GELUGRAD,
                                     colsum_part=dh_part.get(id(bw)))
                        else:
                            dhid = _dgrad(dY[r0:r1], w_t(bw, 0))
                            hip.quickgelu_bwd(L["h"][r0:r1], dhid, dh[r0:r1])
                    del hid
                    for r0, r1, bw in groups:
                        p = names[id(bw)]
                        wide_wgrad(p + ".mlp.c_fc.weight", dh[r0:r1], L["lno2"][r0:r1], r1 - r0, (4 * D, D))
                        if id(bw) in dh_part:
                            set_bias(p + ".mlp.c_fc.bias", dh_part[id(bw)])
                        else:
                            grads[p + ".mlp.c_fc.bias"] = gradgemm.on_lane(lambda a=dh[r0:r1]: hip.colsum(a), dh)
                        _dgrad(dh[r0:r1], w_t(bw, 1), dlno[r0:r1])
                    del dh
                    # attention half.  dX behind the ln_2 backward is out_proj's output gradient: its bf16 copy and bias sums leave with that pass
                    dY2 = torch.empty(M, D, dtype=BF, device=dev)          # not dY again: the lane stream may still read it (c_proj wgrad)
                    bsum = ln_bwd_segments(L["x_mid"], dlno, segs, groups, r_lo, "ln2", i, dY2 if fuse_cast else None)
                    if bsum is None:
                        bsum = cast_with_bias_sums(dX, dY2, r_lo, groups)
                    dao = torch.empty(M, D, dtype=BF, device=dev)
                    dqkv = torch.empty(M, 3 * D, dtype=BF, device=dev)      # attention_bwd writes every row of the towers that ran
                    for r0, r1, bw in groups:
                        p = names[id(bw)]
                        wide_wgrad(p + ".attn.out_proj.weight", dY2[r0:r1], L["ao"][r0:r1], r1 - r0, (D, D))
                        set_bias(p + ".attn.out_proj.bias", bsum[id(bw)])
                        _dgrad(dY2[r0:r1], w_t(bw, 2), dao[r0:r1])
                # the attention backward also leaves every sample's token sums of its dqkv rows (in_proj's bias gradient = their
                # sum over the samples: 1 024 x 3 D fp32 to fold instead of a second pass over dqkv [M, 3 D]); the query-blocked
                # form of the long sequences does not carry them
                qpart = None
                if e.Lv <= 96 and (sv["cap"] is not None or e.Lt <= 96):
                    qpart = torch.empty(Bi + Bt, 3 * D, dtype=F32, device=dev)
                if e.vblk[i] is not None:
                    hip.attention_bwd(L["qkv"][:Mv], L["ao"][:Mv], dao[:Mv], dqkv[:Mv], Bi, e.Lv, e.heads, False,
                                      colsum_part=qpart[:Bi] if qpart is not None else None)
                if sv["cap"] is not None:
                    hip.attention_bwd_varlen(L["qkv"][Mv:M], L["ao"][Mv:M], dao[Mv:M], dqkv[Mv:M], sv["cap"].cu, Bt, sv["Lmax"],
                                             e.heads, True, pad_rows=sv["pad"], colsum_part=qpart[Bi:] if qpart is not None else None)
                else:
                    hip.attention_bwd(L["qkv"][Mv:M], L["ao"][Mv:M], dao[Mv:M], dqkv[Mv:M], Bt, e.Lt, e.heads, True,
                                      colsum_part=qpart[Bi:] if qpart is not None else None)
                for r0, r1, bw in groups:
                    p = names[id(bw)]
                    def unscale_q(g):                                                          # packed q rows = 64^-0.5 * W_q
                        g[:D] *= 0.125
                        return g
                    wide_wgrad(p + ".attn.in_proj_weight", dqkv[r0:r1], L["lno1"][r0:r1], r1 - r0, (3 * D, D),
                               post=unscale_q)                                          # wrt the PACKED weight, scaled back
                    if qpart is not None:
                        fold_into(qpart[(0 if r0 < Mv else Bi):(Bi + Bt if r1 > Mv else Bi)],
                                  lambda r, k=p + ".attn.in_proj_bias": grads.__setitem__(k, r), scale_n=D, scale=0.125)
                    else:
                        def bias_q(a=dqkv[r0:r1]):
                            g = hip.colsum(a)
                            g[:D] *= 0.125
                            return g
                        grads[p + ".attn.in_proj_bias"] = gradgemm.on_lane(bias_q, dqkv)
                    _dgrad(dqkv[r0:r1], w_t(bw, 3), dlno[r0:r1])
                # dX behind the ln_1 backward is the output gradient of the block below's c_proj -- unless a lateral adapter's token
                # path adds to the image rows first, or the block below runs other rows
                below = sv["layers"][i - 1] if i > 0 else None
                same = (below is not None and L["adapter"] is None and below["r_lo"] == r_lo and
                        [(a, b_) for a, b_, _ in below["segs"]] == [(a, b_) for a, b_, _ in segs] and
                        [(a, b_) for a, b_, _ in below["groups"]] == [(a, b_) for a, b_, _ in groups])
                dY_below = torch.empty(M, D, dtype=BF, device=dev) if (fuse_cast and same) else None
                bs = ln_bwd_segments(L["x_in"], dlno, segs, groups, r_lo, "ln1", i, dY_below)
                if bs is not None:
                    # keyed by THIS block's weight sets; the block below holds its own: same row ranges, so map by range
                    by_range = {(g0, g1): bs[id(bw)] for g0, g1, bw in groups}
                    carry = (dY_below, {id(bw): by_range[(g0, g1)] for g0, g1, bw in below["groups"]})
                if L["adapter"] is not None:                                                   # token path of the lateral adapter
                    ad = L["adapter"]
                    a = e.adapters[ad["j"]]
                    dsum = torch.empty(Mv, D, dtype=F32, device=dev)
                    part, _ = hip.layernorm_bwd(ad["sum"], dX[:Mv], a["ln"].g, dsum, Mv, accumulate=False, fold=False)
                    ln_param_grads(part, f"visual.transformer.parallel_lateral_adapter.{ad['j']}.ln_adapt")
                    if self.bn == "batch":                               # adapter convs + parallel stage j (train_conv.py)
                        dgrid, dww = conv.adapter(grads, ad["j"], dsum, ad["x_pre"])
                    else:
                        # (moving this chain -- bandwidth-bound, nothing on the critical path reads its products -- to the lane
                        #  stream was measured: 88.5-88.9 -> 91.0-91.1 ms per step, its kernels take CUs from the dgrad GEMMs)
                        conv.adapter(grads, ad["j"], dsum, ad["x_pre"])
                        dgrid, dww = dsum, a["dww"]
                    hip.adapter_dx(dgrid, dww, dX[:Mv], Bi, e.Lv, e.g, e.usecls)
                sv["layers"][i] = None                                                         # free the layer's activations
            # ---- fronts: text embedding, image cls / positional embeddings, ln_pre
            # the text rows of dX are final here and nothing below reads the result: the scatter-add (0.67 ms of atomics at
            # batch 512) runs on the lane, beside the stem's backward
            dX_text, tok_ids, sv_cap, n_live = dX[Mv:M], sv["tok"], sv["cap"], sv["Mt_live"]

            ne = e.emb.numel()

            def embed_bwd():
                flat = torch.zeros(ne + e.Lt * D, dtype=F32, device=dev)                     # one tensor: on_lane's contract
                # the positional embedding's gradient is a sum over the batch: a column sum in a fixed order (bitwise repeatable),
                # not the kernel's atomics; the token embedding's scatter-add stays atomic (captions share ids)
                if sv_cap is not None:
                    # packed rows: scatter-add over the live rows, positional sums over the captions that have the position
                    hip.embed_tokens_bwd_packed(tok_ids, dX_text[:n_live], sv_cap.cu, flat[:ne].view_as(e.emb), flat[ne:].view(e.Lt, D))
                elif dX_text.is_contiguous():
                    hip.embed_tokens_bwd(tok_ids, dX_text, flat[:ne].view_as(e.emb), None)
                    hip.colsum(dX_text.view(Bt, e.Lt * D), out=flat[ne:])
                else:
                    hip.embed_tokens_bwd(tok_ids, dX_text, flat[:ne].view_as(e.emb), flat[ne:].view(e.Lt, D))
                return flat
            both = gradgemm.on_lane(embed_bwd, dX_text, tok_ids, *([sv_cap.cu] if sv_cap is not None else []))
            grads["token_embedding.weight"], grads["positional_embedding"] = both[:ne].view_as(e.emb), both[ne:].view(e.Lt, D)
            dtok = torch.empty(Mv, D, dtype=F32, device=dev)
            part, _ = hip.layernorm_bwd(sv["tok_pre"], dX[:Mv], e.ln_pre.g, dtok, Mv, accumulate=False, fold=False)
            ln_param_grads(part, "visual.ln_pre")
            dvpos = hip.colsum(dtok.view(Bi, e.Lv * D)).view(e.Lv, D)                           # sum over the batch
            grads["visual.positional_embedding"] = dvpos
            grads["visual.class_embedding"] = dvpos[0].clone()
            folds.run()                                      # every deferred column sum of the transformer's backward, two launches
            conv.stem(grads, dtok)
            flush_wgrads()
            gradgemm.join(dev)                                   # the gradients queued on the lane stream
            sv["w"].pop("held", None)
            self.saved = None
            return reducer.finish(clone=clone) if reducer is not None else dict(grads)

    # ------------------------------------------------------------------ optimizer
    def set_epoch(self, epoch):
        """Apply the yaml's learning-rate schedule (self.schedule, train.from_config) for `epoch` to both parameter groups'
        base rates; the next step() re-points the optimizer table's rates.  No-op without a schedule."""
        sch = getattr(self, "schedule", None)
        if sch is None:
            return
        if not hasattr(self, "_base_lr"):
            self._base_lr = (self.lr, self.lr_share)
        base, base_sh = self._base_lr
        self.lr = sch.lr_at(base, epoch)
        self.lr_share = None if base_sh is None else sch.lr_at(base_sh, epoch)

    def param_groups(self):
        """(name, parameter, lr, weight_decay) for every parameter: module-level param_groups() with this step's settings."""
        return param_groups(self.model, self.lr, self.lr_share, self.wd, self.wd_share, self.without_wd)

    def _packed_destinations(self):
        """{id(parameter): [(first element, element count, packed tensor, scale)]}: the engine's copies of the transformer
        blocks' projection tensors that are plain (scaled) casts of a parameter -- AdamW writes them from the same kernel as
        the parameter itself (msclip_adamw_tensor.pk).  in_proj: the q rows carry head_dim^-0.5 (packing.qkv_weights)."""
        e = self.eng
        out = {}
        for blk in {id(b["w"]): b["w"] for b in list(e.tblk) + [b for b in e.vblk if b is not None]}.values():
            a, mlp = blk.blk.attn, blk.blk.mlp
            d = a.in_proj_weight.shape[1]
            out[id(a.in_proj_weight)] = [(0, d * d, blk.wqkv.view(-1)[:d * d], blk.qscale),
                                         (d * d, 2 * d * d, blk.wqkv.view(-1)[d * d:], 1.0)]
            out[id(a.in_proj_bias)] = [(0, d, blk.bqkv[:d], blk.qscale), (d, 2 * d, blk.bqkv[d:], 1.0)]
            out[id(a.out_proj.weight)] = [(0, blk.wo.numel(), blk.wo.view(-1), 1.0)]
            out[id(mlp.c_fc.weight)] = [(0, blk.wfc.numel(), blk.wfc.view(-1), 1.0)]
            out[id(mlp.c_proj.weight)] = [(0, blk.wpr.numel(), blk.wpr.view(-1), 1.0)]
        return out

    def _adamw_plan(self, grads):
        """The tensor table of the optimizer launch: built once, per step only the gradient addresses are re-pointed;
        rebuilt when the set of gradients changes, the optimizer state is replaced or the engine has re-packed (new copies)."""
        pk_ok = not self.eng.fp8
        sig = (id(self.eng.tblk[0]["w"].wqkv), pk_ok, id(self.state), self.eng.tensor_identity())
        plan = getattr(self, "_plan", None)
        if plan is not None and plan.sig == sig and len(grads) == plan.ngrads and \
                all(k in grads and grads[k].dtype == F32 and grads[k].numel() == n for k, n in plan.names.items()):
            # same tensors, this step's gradients (fresh allocations on one GPU, the bucket slots at N > 1); the few that
            # arrive as permuted views (depthwise adapter weights) are made contiguous, alive until the launch is queued
            flat = {k: (grads[k] if grads[k].is_contiguous() else grads[k].contiguous()) for k in plan.names}
            plan.hold = flat
            plan.set_grads([flat[k].data_ptr() + 4 * lo for k, lo in plan.pieces])
            if plan.rates_for != (self.lr, self.lr_share, self.wd, self.wd_share):
                groups = {k: (lr, wd) for k, _, lr, wd in self.param_groups()}
                plan.set_rates([groups[k] for k, _ in plan.pieces])
                plan.rates_for = (self.lr, self.lr_share, self.wd, self.wd_share)
            return plan
        dests = self._packed_destinations() if pk_ok else {}
        items, pieces, names, hold = [], [], {}, []
        for k, p, lr, wd in self.param_groups():
            g = grads.get(k)
            if g is None:
                continue
            g = g.reshape(p.shape).contiguous().view(-1)
            hold.append(g)
            st = self.state.get(k)
            if st is None:
                st = self.state[k] = (torch.zeros_like(p), torch.zeros_like(p))
            pf, mf, vf = p.data.view(-1), st[0].view(-1), st[1].view(-1)
            names[k] = p.numel()
            for lo, n, pk, scale in dests.get(id(p), [(0, p.numel(), None, 1.0)]):
                items.append((pf[lo:lo + n], g[lo:lo + n], mf[lo:lo + n], vf[lo:lo + n], lr, wd, pk, scale))
                pieces.append((k, lo))
        plan = hip.AdamwPlan(items)
        plan.sig, plan.pieces, plan.names, plan.ngrads = sig, pieces, names, len(grads)
        plan.rates_for = (self.lr, self.lr_share, self.wd, self.wd_share)
        plan.packs = bool(dests)
        plan.hold = hold
        self._plan = plan
        return plan

    @hip.off_default_stream
    def step(self, grads, world_average=False):
        """AdamW on the module's fp32 parameters (msclip_adamw_multi), which also writes the engine's bf16 copies of the
        transformer blocks' projections; the engine re-packs the rest (conv side, heads).  backward() already returns
        rank-averaged gradients; world_average=True averages here instead, tensor by tensor (for gradients produced with
        backward(reduce=False))."""
        if self.lr is None:
            raise ValueError("TrainStep.step() needs a learning rate: TrainStep(model, lr=...) or train.from_config(model, config)")
        if hasattr(grads, "check_fresh"):
            grads.check_fresh()
        self.steps += 1
        with torch.no_grad():
            if world_average:
                world_average_(grads)
            plan = self._adamw_plan(grads)
            plan.run(self.betas[0], self.betas[1], self.eps, self.steps)
            plan.hold = None
        if plan.packs:
            self.eng.repack_after_optimizer()
        else:
            self.eng.refresh(force=True)


def world_average_(grads):
    """In-place rank mean of a gradient dict produced with backward(reduce=False) (what step(world_average=True) runs): one
    all-reduce per tensor.  Non-contiguous entries (the conv side's depthwise weight gradients are transposed views,
    train_conv.py) are replaced by dense copies first: NCCL / gloo reject strided tensors."""
    if C.comm.world_size <= 1:
        return grads
    for k in list(grads):
        g = grads[k]
        if not g.is_contiguous():
            g = grads[k] = g.contiguous()
        dist.all_reduce(g)
        g /= C.comm.world_size
    return grads


def param_groups(model, lr, lr_share, wd, wd_share, without_wd=("bn", "bias", "ln")):
    """(name, parameter, lr, weight_decay) for every parameter of the slice, following the reference's yaml: shared
    attention / MLP tensors use CUSTOM.LR_SHARE / WD_SHARE, everything else TRAIN.LR / TRAIN.WD; no decay for the
    TRAIN.WITHOUT_WD_LIST keywords -- 'bn' = parameters of BatchNorm modules (whatever their name: the stem's
    `downsample.1` is one), 'ln' = parameters of LayerNorm modules, 'bias' = names ending in 'bias' -- and for every
    name that CONTAINS an entry of model.no_weight_decay() (M.py:2950-2956: 'positional_embedding' covers
    visual.positional_embedding, 'token_embedding' covers token_embedding.weight)."""
    m = model
    norm_params = set()
    for mod in m.modules():
        if (isinstance(mod, torch.nn.BatchNorm2d) and "bn" in without_wd) or \
           (isinstance(mod, torch.nn.LayerNorm) and "ln" in without_wd):
            norm_params.update(id(p) for p in mod.parameters(recurse=False))
    shared = set()
    if m.share_from_layer is not None:
        for i in range(max(m.share_from_layer, 1), len(m.visual.transformer.resblocks)):
            shared.update(f"visual.transformer.resblocks.{i}.{leaf}" for leaf in _LEAVES)
    nodecay = set(m.no_weight_decay())
    params = dict(m.named_parameters())
    out = []
    for k, p in params.items():
        plr = lr_share if (k in shared and lr_share is not None) else lr
        pwd = wd_share if (k in shared and wd_share is not None) else wd
        if id(p) in norm_params or ("bias" in without_wd and k.endswith("bias")) or any(t in k for t in nodecay):
            pwd = 0.0
        out.append((k, p, plr, pwd))
    return out


def _optimizer_state_dict(ts):
    """TrainStep's AdamW state in torch.optim.AdamW's state_dict layout (parameter indices in named_parameters order, one
    param_group per distinct (lr, weight_decay)): what the reference's `save_checkpoint_on_master` stores under
    'optimizer' (lib/utils/utils.py:177-183)."""
    groups, index = {}, {}
    state = {}
    for i, (k, p, lr, wd) in enumerate(ts.param_groups()):
        index[k] = i
        groups.setdefault((lr, wd), []).append(i)
        st = ts.state.get(k)
        if st is not None:
            state[i] = {"step": torch.tensor(float(ts.steps)), "exp_avg": st[0].detach().cpu(), "exp_avg_sq": st[1].detach().cpu()}
    pgs = [{"lr": lr, "weight_decay": wd, "betas": tuple(ts.betas), "eps": ts.eps, "amsgrad": False, "params": idx}
           for (lr, wd), idx in groups.items()]
    return {"state": state, "param_groups": pgs, "msclip": {"steps": ts.steps, "bn": ts.bn, "names": list(index)}}


def save_checkpoint(model, ts, path, step, model_name="", perf=0.0):
    """The reference's resumable checkpoint dict (lib/utils/utils.py:157-200): 'step', 'model', 'state_dict', 'perf',
    'optimizer'.  Rank 0 only under N > 1."""
    if not C.comm.is_main_process():
        return
    torch.save({"step": step + 1, "model": model_name, "perf": perf,
                "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "optimizer": _optimizer_state_dict(ts)}, path)


def resume_checkpoint(model, ts, path):
    """-> the step to continue from.  Loads the module (strict, aliases checked), the AdamW moments and the step count."""
    from .checkpoint import check_aliases, extract_state_dict
    obj = torch.load(path, map_location="cpu", weights_only=False)
    sd = extract_state_dict(obj)
    model.load_state_dict(sd, strict=True)
    check_aliases(model, sd)
    opt = obj["optimizer"]
    names = opt["msclip"]["names"]
    params = dict(model.named_parameters())
    ts.state = {}
    for i, st in opt["state"].items():
        p = params[names[int(i)]]
        ts.state[names[int(i)]] = (st["exp_avg"].to(p.device).contiguous(), st["exp_avg_sq"].to(p.device).contiguous())
    ts.steps = int(opt["msclip"]["steps"])
    ts._plan = None                                   # the cached optimizer table points at the moments just replaced
    ts.eng.refresh(force=True)
    return int(obj.get("step", ts.steps))


def from_config(model, config, bn="batch"):
    """TrainStep with the reference yaml's optimizer block (experiments/model/b32.yaml:32-52 + the msclips overlays):
    TRAIN.OPTIMIZER (only adamW is implemented: anything else raises), TRAIN.LR / WD / WITHOUT_WD_LIST, TRAIN.OPTIMIZER_ARGS
    (betas / eps; absent => torch.optim.AdamW's defaults, what `AdamW(params, lr=..., weight_decay=..., **{})` gives),
    CUSTOM.LR_SHARE / WD_SHARE for the modality-shared tensors (already scaled with the world size by update_config,
    lib/config/default.py:299-304).
    bn = "batch" (default): train-mode BatchNorm as the reference's modules run in train(); "frozen": running statistics."""
    ts = TrainStep(model, bn=bn, **optimizer_settings(config))
    ts.schedule = lr_schedule(config)            # TRAIN.LR_SCHEDULER of the yaml (None when the config has none)
    return ts


class CosineSchedule:
    """TRAIN.LR_SCHEDULER {METHOD: 'timm', ARGS: {sched: 'cosine', ...}} of the reference yamls (experiments/model/b32.yaml:40-48;
    lib/config/default.py:306-308 adds ARGS.epochs = TRAIN.END_EPOCH and hands ARGS to timm's create_scheduler in the unreleased
    trainer).  timm's CosineLRScheduler stepped once per EPOCH (t_in_epochs), one cycle, no warm-up prefix:
        t <  warmup_epochs:  lr = warmup_lr + t * (base - warmup_lr) / warmup_epochs
        t <  epochs:         lr = min_lr + 0.5 * (base - min_lr) * (1 + cos(pi * t / epochs))
        t >= epochs:         lr = min_lr        (the cooldown_epochs run at min_lr; the run lasts epochs + cooldown_epochs)
    applied to every parameter group's own base rate (TRAIN.LR for the modality-specific tensors, CUSTOM.LR_SHARE for the shared)."""

    def __init__(self, epochs, warmup_epochs=0, warmup_lr=0.0, min_lr=0.0, cooldown_epochs=0, decay_rate=0.1):
        self.epochs, self.warmup_epochs, self.warmup_lr, self.min_lr = int(epochs), int(warmup_epochs), float(warmup_lr), float(min_lr)
        self.cooldown_epochs, self.decay_rate = int(cooldown_epochs), float(decay_rate)      # (decay_rate only acts on later cycles: one cycle here)
        self.total_epochs = self.epochs + self.cooldown_epochs

    def lr_at(self, base, epoch):
        import math
        t = int(epoch)
        if t < self.warmup_epochs:
            return self.warmup_lr + t * (base - self.warmup_lr) / self.warmup_epochs
        if t < self.epochs:
            return self.min_lr + 0.5 * (base - self.min_lr) * (1.0 + math.cos(math.pi * t / self.epochs))
        return self.min_lr


def lr_schedule(config):
    """The scheduler block of a reference config (needs no GPU); None when TRAIN.LR_SCHEDULER is absent."""
    tr = config.TRAIN
    sch = tr.get("LR_SCHEDULER", None)
    if not sch or not sch.get("METHOD", None):
        return None
    if str(sch.METHOD) != "timm":
        raise NotImplementedError(f"TRAIN.LR_SCHEDULER.METHOD = {sch.METHOD!r}: the released configs use 'timm' (cosine)")
    a = dict(sch.get("ARGS", None) or {})
    if str(a.get("sched", "cosine")) != "cosine":
        raise NotImplementedError(f"TRAIN.LR_SCHEDULER.ARGS.sched = {a.get('sched')!r}: only timm's cosine schedule is implemented")
    # the reference's update_config sets ARGS.epochs = TRAIN.END_EPOCH unconditionally (lib/config/default.py:306-311): the yaml's
    # END_EPOCH wins over an `epochs` key under ARGS
    epochs = int(tr.get("END_EPOCH", 0) or 0) or int(a.get("epochs", 0) or 0)
    warm = int(a.get("warmup_epochs", 0) or 0)
    if epochs <= 0 or epochs <= warm:
        raise ValueError(f"TRAIN.LR_SCHEDULER: the cosine schedule needs TRAIN.END_EPOCH ({epochs}) > ARGS.warmup_epochs ({warm})")
    return CosineSchedule(epochs=epochs, warmup_epochs=warm,
                          warmup_lr=a.get("warmup_lr", 0.0), min_lr=a.get("min_lr", 0.0),
                          cooldown_epochs=a.get("cooldown_epochs", 0), decay_rate=a.get("decay_rate", 0.1))


def optimizer_settings(config):
    """The optimizer block of a reference config as TrainStep keyword arguments (needs no GPU)."""
    tr, cu = config.TRAIN, config.CUSTOM
    opt = str(tr.get("OPTIMIZER", "sgd"))
    if opt.lower() != "adamw":
        raise NotImplementedError(f"TRAIN.OPTIMIZER = {opt!r}: this build implements adamW only (the released MS-CLIP-S "
                                  "configs' optimizer, experiments/model/b32.yaml:48)")
    oa = dict(tr.get("OPTIMIZER_ARGS", None) or {})
    betas = tuple(oa.pop("betas", (0.9, 0.999)))
    eps = float(oa.pop("eps", 1e-8))
    oa.pop("lr", None)
    if oa:
        raise NotImplementedError(f"TRAIN.OPTIMIZER_ARGS keys {sorted(oa)} are not implemented")
    return dict(lr=tr.LR, lr_share=cu.get("LR_SHARE", None) or None, wd=tr.WD, wd_share=cu.get("WD_SHARE", None) or None,
                betas=betas, eps=eps, without_wd=tuple(tr.get("WITHOUT_WD_LIST", ()) or ()))
