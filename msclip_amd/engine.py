"""Execution engine of the MS-CLIP-S hot path on one MI355X.

Packs the module's parameters once (bf16 GEMM weights, folded BatchNorm,
chunk tables; msclip_amd.packing) and drives the HIP kernels of
libmsclip_hip.so through msclip_amd.hip.  Data layout in HBM:

* token stream: ONE fp32 residual matrix X[Mv + Mt, D], image tokens of all
  samples first (sample-major, L_img rows each), then text tokens.  Layers whose
  attention/MLP tensors are shared between the towers (reference M.py:2786-2830)
  run ONE GEMM over all Mv + Mt rows per projection; LayerNorms and attention
  stay per-modality (own gamma/beta, own sequence length / causal flag).
* GEMM operands/outputs bf16 (LN output, QKV, attention output, MLP hidden),
  accumulation and residual stream fp32.
* convolutional activations NHWC bf16, so a conv is the gathering GEMM.

Control flow follows Transformer.forward (M.py:2388-2459): stem -> tokens ->
for i in 1..11: [parallel stage + lateral adapter before blocks 2,4,6,8,10] ->
block i; text block 0 is text-only.
"""
import os

import torch
import torch.distributed as dist

from . import comm as C
from . import hip
from . import packing as P
from .options import EngineOptions


_TEXT0 = {}
_MASKED = {}


def _text0_stream(device):
    """The side stream of the text front + text block 0 (one per device)."""
    s = _TEXT0.get(device)
    if s is None:
        s = _TEXT0[device] = torch.cuda.Stream(device=device)
    return s


class Captions:
    """A caption batch staged for the engine (Engine.stage_captions): the int64 token ids [B, L] plus, computed on the device
    when the batch was staged, every caption's live length n_b = argmax + 1, the exclusive prefix sums `cu` (+ total, +
    maximum) and the EOT row of each caption relative to the packed text segment; the two host-side numbers the engine needs
    (total live rows, longest caption) travel through pinned memory behind `event`.  Staging a batch while the previous one
    is still being computed (an input pipeline's prefetch stage) means the engine never waits for them."""

    def __init__(self, tok, length, cu, eot, host, event, release=None):
        self.tok, self.len, self.cu, self.eot, self._host, self._event = tok, length, cu, eot, host, event
        self._totals = None
        self._release = release              # gives the pinned read-back slot back to the engine's pool once it has been read
        self.shape = tok.shape

    def __del__(self):
        # a batch that was staged but never consumed: its slot may only be reused once the copy into it has landed
        if self._host is not None and self._release is not None:
            try:
                self._event.synchronize()
                self._release(self._host)
            except Exception:
                pass

    def ready(self):
        """True when totals() would not block (the 8-byte read-back has landed)."""
        return self._totals is not None or self._event.query()

    def totals(self):
        """(total live rows, longest caption): the one host read, taken once."""
        if self._totals is None:
            self._event.synchronize()
            self._totals = (int(self._host[0]), int(self._host[1]))
            if self._release is not None:
                self._release(self._host)
            self._host = None
        return self._totals


class _BlockW:
    """Packed weights of one ResidualAttentionBlock's shareable part."""

    def __init__(self, blk, heads, fp8=False, fp8_qkv=False):
        self.blk = blk                               # the module these copies were made from (train.TrainStep maps parameters to them)
        self.qscale = float(blk.attn.in_proj_weight.shape[1] // heads) ** -0.5
        self.wqkv, self.bqkv = P.qkv_weights(blk.attn.in_proj_weight.detach(), blk.attn.in_proj_bias.detach(), heads)
        if fp8:
            # MODEL.SPEC.PRECISION fp8 (BASELINE config C5): the LayerNorm-fed projections' weights as OCP e4m3 with one
            # scale per output channel, quantised from the fp32 parameters (q rows carry 64^-0.5 like the bf16 copy)
            if fp8_qkv:
                d = blk.attn.in_proj_weight.shape[1]
                wq = blk.attn.in_proj_weight.detach().float().clone()
                wq[:d] *= self.qscale
                self.wqkv_q, self.wqkv_s = hip.quantize_rows_f8(wq)
            self.wfc_q, self.wfc_s = hip.quantize_rows_f8(blk.mlp.c_fc.weight.detach())
            self.wpr_q, self.wpr_s = hip.quantize_rows_f8(blk.mlp.c_proj.weight.detach())
            # static scale of the MLP hidden matrix (c_fc's e4m3 output = c_proj's operand): calibrated on the first batch
            # this layer sees (Engine._mlp_f8), None until then
            self.hid_scale = None
            self.wpr_cs = None
        bf = torch.bfloat16
        self.wo = blk.attn.out_proj.weight.detach().to(bf).contiguous()
        self.bo = blk.attn.out_proj.bias.detach().float().contiguous()
        self.wfc = blk.mlp.c_fc.weight.detach().to(bf).contiguous()
        self.bfc = blk.mlp.c_fc.bias.detach().float().contiguous()
        self.wpr = blk.mlp.c_proj.weight.detach().to(bf).contiguous()
        self.bpr = blk.mlp.c_proj.bias.detach().float().contiguous()


class _LN:
    def __init__(self, ln):
        self.g = ln.weight.detach().float().contiguous()
        self.b = ln.bias.detach().float().contiguous()


class Engine:
    def __init__(self, model):
        hip.require_gpu()
        ref = model.visual.positional_embedding
        if not ref.is_cuda:
            raise hip.HipUnavailable("model parameters are on the CPU: call model.cuda() (there is no CPU path)")
        if ref.dtype != torch.float32:
            raise NotImplementedError("keep the module in fp32 (checkpoint ABI); the engine makes its own bf16 copies")
        self.dev = ref.device
        self.model = model
        self.opt = EngineOptions.from_env()      # every run-time switch, read once (msclip_amd/options.py); tests replace fields
        self._ws = {}
        self._calib = None              # calibrate_fp8() in progress: {id(_BlockW): (block, [device amax, ...])}
        self._rec = None                # the launch plan (hip.Plan) that is recording this call's launches, if any
        self.last_plan = None
        self._fp8_saved = {}            # calibrated hidden scales, restored into the fresh _BlockW objects of a re-pack
        self._fp8_warned = False
        self.force_unfused = False      # the training step runs the conv side layer by layer: every map stays in the workspace
        with torch.cuda.device(self.dev), torch.no_grad():
            self._pack(model)
        self._stamp = self._fingerprint()

    def _module_tensors(self):
        """Every parameter / buffer object of the module tree, by a direct walk of the modules' own dicts (shared modules are
        visited once per path: harmless for a fingerprint).  Module.parameters() / .buffers() build the same set through
        named_modules + de-duplicating generators: 1.5 ms per call for this model's ~600 tensors against ~0.15 ms here, and
        the training step asks four times per step on a host the GPU is waiting for at the step boundary."""
        out, stack = [], [self.model]
        while stack:
            m = stack.pop()
            out.extend(m._parameters.values())
            out.extend(m._buffers.values())
            stack.extend(m._modules.values())
        return [t for t in out if t is not None]

    def _fingerprint(self):
        """Identity + in-place version of every parameter and buffer the packed copies were made from.  Any mutation
        the module hooks cannot see (submodule load_state_dict, param.copy_/fill_/clamp_, an optimizer step, a
        re-assigned .data) changes it; run() then re-packs instead of silently serving stale weights."""
        v = 0
        for t in self._module_tensors():
            v = (v * 1000003 + t._version * 31 + t.data_ptr()) & 0xFFFFFFFFFFFF
        return v

    def refresh(self, force=False):
        """Re-pack the weights if the module's tensors changed since the last pack (cheap check: ~0.1 ms)."""
        stamp = self._fingerprint()
        if force or stamp != self._stamp:
            ref = self.model.visual.positional_embedding
            if not ref.is_cuda or ref.device != self.dev:
                raise hip.HipUnavailable("model parameters left the engine's device: rebuild the engine (model.engine())")
            with torch.cuda.device(self.dev), torch.no_grad():
                self._sdv = None
                self._pack(self.model)
            self._stamp = stamp
            self._ws = {k: w for k, w in self._ws.items() if not (isinstance(k, tuple) and k and k[0] == "graph")}
            self.drop_plans()                        # (launch tables hold the old packed tensors' addresses)
            return True
        return False

    def repack_after_optimizer(self):
        """The re-pack behind an in-place optimizer step whose kernel has already written the transformer blocks' bf16 /
        scaled copies itself (msclip_adamw_multi's `pk` outputs, train.TrainStep.step): the conv side (BatchNorm folds, NHWC
        weight matrices), the heads and the logit scale, from the cached state-dict views -- not the 0.8 ms state_dict()
        walk and the ~100 per-tensor casts of the blocks on a host that the GPU is waiting for.  Falls back to the full
        re-pack for fp8 models (their e4m3 copies are quantised per row on the host side of the pack)."""
        if self.fp8 or getattr(self, "_sdv", None) is None or self.tensor_identity() != self._sdv_ident:
            return self.refresh(force=True)          # (a re-assigned parameter: the blocks' copies / aliases are of the old tensor)
        with torch.cuda.device(self.dev), torch.no_grad():
            plan = getattr(self, "_pack_plan", None)
            if plan is not None and not self.patch and self.opt.repack_table:
                # round 5: every derived conv-side tensor rewritten in place by ONE launch (msclip_pack_weights), the two
                # projection heads by a transposing copy each; everything else the launches read is a view of the parameters
                plan.run()
                sd = self._sdv
                self.w_vproj.copy_(sd["visual.proj"].t())
                self.w_tproj.copy_(sd["text_projection"].t())
                self._foldw = {}                     # (the LayerNorm fold's gamma-folded weights are of the old values)
                self._ls_host.copy_(self.model.logit_scale.detach().exp().reshape(1), non_blocking=True)
                self._ls_event = torch.cuda.Event()
                self._ls_event.record(torch.cuda.current_stream(self.dev))
            else:
                self._pack(self.model, blocks=False)
        self._stamp = self._fingerprint()
        self._ws = {k: w for k, w in self._ws.items() if not (isinstance(k, tuple) and k and k[0] == "graph")}
        self.drop_plans()
        return True

    def state_views(self):
        """{state_dict key: detached tensor} of the module, made once per full re-pack: the tensors are the module's own
        storage, so in-place updates (an optimizer step, load_state_dict) show through; a re-assigned parameter changes the
        fingerprint and with it this cache."""
        ident = self.tensor_identity()
        if getattr(self, "_sdv", None) is None or self._sdv_ident != ident:
            self._sdv = {k: t.detach() for k, t in self.model.state_dict().items()}
            self._sdv_ident = ident
        return self._sdv

    def tensor_identity(self):
        """Which tensor OBJECTS the module holds (a parameter that has been re-assigned is another object): what cached
        views of them and cached optimizer tables are valid for."""
        v = 0
        for t in self._module_tensors():
            v = (v * 1000003 + id(t)) & 0xFFFFFFFFFFFF
        return v

    # ------------------------------------------------------------------ packing
    def _pack(self, m, blocks=True):
        dev = self.dev
        v, vt = m.visual, m.visual.transformer
        sd = dict(self.state_views())                # a copy: bn_fold_all parks its folds in it
        P.bn_fold_all(sd)
        self.D = m.transformer_width
        self.E = m.embed_dim
        self.heads = m.heads
        self.Lv = v.sequence_length
        self.g = v.input_resolution // v.patch_size
        self.Lt = m.context_length
        self.S = v.input_resolution
        self.n_layers = len(m.transformer.resblocks)
        self.fp8 = getattr(m, "precision", "bf16").startswith("fp8")          # c_fc / c_proj on the fp8 MFMA
        self.fp8_qkv = getattr(m, "precision", "bf16") == "fp8-qkv"           # ... and in_proj
        if self.fp8 and self.D % 128:
            raise NotImplementedError("PRECISION fp8 needs a width that is a multiple of 128 (one K-tile of the MX MFMA)")
        assert len(vt.resblocks) == self.n_layers, "vision and text depth must match for the batched layer loop"
        # exp(logit_scale) is a HOST scalar for the logits kernels (GEMM alpha / LSE scale).  Reading it here would block the
        # host until everything queued before the re-pack has finished -- after an optimizer step that is the whole
        # training step, and the next step's launches would start against an empty GPU queue.  So: an asynchronous copy into
        # pinned memory now, the wait when the value is first used (the logits stage, at the end of a forward).
        self._ls_host = torch.empty(1, dtype=torch.float32).pin_memory()
        self._ls_host.copy_(m.logit_scale.detach().exp().reshape(1), non_blocking=True)
        self._ls_event = torch.cuda.Event()
        self._ls_event.record(torch.cuda.current_stream(dev))

        self.patch = hasattr(v, "conv1")               # plain patch-conv stem (EARLY_CONV off, M.py:2502-2508): BASELINE config C5's ViT-L/14
        self.cls = sd["visual.class_embedding"].float().contiguous()
        self.vpos = sd["visual.positional_embedding"].float().contiguous()
        self.ln_pre, self.ln_post = _LN(v.ln_pre), _LN(v.ln_post)
        self.w_vproj = sd["visual.proj"].t().to(torch.bfloat16).contiguous()          # [E, D]
        if self.patch:
            # conv1 (kernel == stride == patch, no bias) is a dense GEMM over the patch matrix msclip_patchify builds:
            # K index = (c, kh, kw) as the weight lies in memory, zero-padded to a multiple of 64
            kp = 3 * v.patch_size * v.patch_size
            self.patch_k = (kp + 63) // 64 * 64
            wp = torch.zeros(self.D, self.patch_k, dtype=torch.bfloat16, device=dev)
            wp[:, :kp] = sd["visual.conv1.weight"].reshape(self.D, kp).to(torch.bfloat16)
            self.w_patch = wp
            self.stem_specs, self.lateral, self.usecls, self.adapters = [], [], False, []
            self.par_specs, self.par_b3r, self.par_hw, self.h1 = [None], [None], [], 0
        else:
            self._pack_conv_side(sd, v, vt, dev)

        # --- transformer blocks (shared tensors packed once)
        cache = {}
        self._foldw = {}                             # gamma-folded projection weights of the LayerNorm fold, built on first use

        def blockw(blk):
            key = (blk.attn.in_proj_weight.data_ptr(), blk.attn.out_proj.weight.data_ptr(),
                   blk.mlp.c_fc.weight.data_ptr(), blk.mlp.c_proj.weight.data_ptr())
            if key not in cache:
                cache[key] = _BlockW(blk, self.heads, self.fp8, self.fp8_qkv)
            return cache[key]

        if blocks:                                   # (False: repack_after_optimizer, the copies are already current)
            self.vblk, self.tblk = [None] * self.n_layers, [None] * self.n_layers
            for i in range(self.n_layers):
                tb = m.transformer.resblocks[i]
                self.tblk[i] = dict(w=blockw(tb), ln1=_LN(tb.ln_1), ln2=_LN(tb.ln_2))
                if i >= 1 or self.patch:                 # (slot 0 is the conv stem unless the patch conv tokenises)
                    vb = vt.resblocks[i]
                    self.vblk[i] = dict(w=blockw(vb), ln1=_LN(vb.ln_1), ln2=_LN(vb.ln_2))
            self.n_packed_blocks = len(cache)
            if self.fp8 and self._fp8_saved:             # a re-pack (optimizer step, load_state_dict) keeps the calibration
                self.load_fp8_state(self._fp8_saved)

        # --- text front / heads
        self.emb = m.token_embedding.weight.detach()
        self.tpos = sd["positional_embedding"].float().contiguous()
        self.ln_final = _LN(m.ln_final)
        self.w_tproj = sd["text_projection"].t().to(torch.bfloat16).contiguous()      # [E, D]

    def _pack_conv_side(self, sd, v, vt, dev):
        """The MS-CLIP-S conv stem (slot 0 of the visual Transformer), the parallel branch and the lateral adapters."""
        # --- stem
        sp = "visual.transformer.resblocks.0"
        self.dual_w, self.dual_b = [t.to(dev) for t in P.stem_dual_weights(sd, sp, "visual.transformer.parallel_branch_v.0")]
        stem = vt.resblocks[0]
        h = (self.S + 2 - 3) // 2 + 1
        self.h1 = h
        self.stem_specs = []
        for i, s in enumerate(stem.strides):
            spec = P.stem_stage(sd, f"{sp}.resnet_stage.conv_{i}", h, s).to(dev)
            self.stem_specs.append(spec)
            h = spec.h_out
        assert h == self.g, f"stem output grid {h} != token grid {self.g}"
        self.w_last = sd[sp + ".last_conv.weight"][:, :, 0, 0].to(torch.bfloat16).contiguous()

        # --- parallel branch + adapters
        self.lateral = list(vt.parallel_lateral_layers)
        self.usecls = vt.t2b_usecls
        self.par_specs, self.par_b3r, self.adapters = [None], [None], []
        h = self.h1
        self.par_hw = [h]
        for j in range(1, 5):
            specs = P.bottleneck(sd, f"visual.transformer.parallel_branch_v.{j}.resnet_stage.conv_0", h,
                                 vt.parallel_strides[j])
            self.par_specs.append([s.to(dev) for s in specs])
            self.par_b3r.append((specs[3].bias + specs[2].bias).contiguous().to(dev))      # conv3 + shortcut bias
            h = specs[3].h_out
            self.par_hw.append(h)
        for j in range(5):
            pool, k, pw, dww, dwb = P.adapter_weights(sd, f"visual.transformer.parallel_lateral_adapter.{j}", self.g)
            assert self.par_hw[j] == self.g * k, (self.par_hw[j], self.g, k)
            self.adapters.append(dict(pool=pool.to(dev), k=k, pw=pw.to(dev), dww=dww.to(dev), dwb=dwb.to(dev),
                                      ln=_LN(vt.parallel_lateral_adapter[j].ln_adapt), C=pool.shape[1]))
        self._pack_plan = self._build_pack_plan(sd, dev)

    def _build_pack_plan(self, sd, dev):
        """The item table of msclip_pack_weights for every derived conv-side tensor packed above (same folds, same layouts, written
        IN PLACE into the tensors the launches read): what repack_after_optimizer runs instead of ~170 ATen launches.  None when a
        source is not plain fp32 device storage (then the tensor-algebra pack stays)."""
        def bn(prefix):
            return tuple(sd[prefix + k] for k in (".weight", ".bias", ".running_mean", ".running_var"))
        try:
            plan = hip.PackPlan(dev)
            sp, pp = "visual.transformer.resblocks.0", "visual.transformer.parallel_branch_v.0"
            c1 = self.dual_w.shape[1] // 2
            plan.add(sd[sp + ".conv1.weight"], self.dual_w, bn=bn(sp + ".bn1"), mode=1, col0=0, ld=2 * c1, bias_out=self.dual_b, bias_mode=1)
            plan.add(sd[pp + ".conv.weight"], self.dual_w, bn=bn(pp + ".bn"), mode=1, col0=c1, ld=2 * c1, bias_out=self.dual_b, bias_mode=1,
                     bias_col0=c1)
            for i, spec in enumerate(self.stem_specs):
                pre = f"{sp}.resnet_stage.conv_{i}"
                plan.add(sd[pre + ".conv1.weight"], spec.weight, bn=bn(pre + ".bn1"), w2=sd[pre + ".downsample.0.weight"],
                         bn2=bn(pre + ".downsample.1"), bias_out=spec.bias, bias_mode=2)
            plan.add(sd[sp + ".last_conv.weight"], self.w_last)
            for j in range(1, 5):
                pre = f"visual.transformer.parallel_branch_v.{j}.resnet_stage.conv_0"
                k1, k2, kr, k3 = self.par_specs[j]
                for spec, conv, b in ((k1, "conv1", "bn1"), (k2, "conv2", "bn2"), (kr, "residual_conv", "residual_bn"), (k3, "conv3", "bn3")):
                    plan.add(sd[f"{pre}.{conv}.weight"], spec.weight, bn=bn(f"{pre}.{b}"), eps=1e-6, bias_out=spec.bias, bias_mode=1)
                plan.add(sd[pre + ".conv3.weight"], None, bn=bn(pre + ".bn3"), eps=1e-6, bn2=bn(pre + ".residual_bn"), eps2=1e-6,
                         bias_out=self.par_b3r[j], bias_mode=2)
            for j, a in enumerate(self.adapters):
                pre = f"visual.transformer.parallel_lateral_adapter.{j}"
                C = a["C"]
                plan.add(sd[pre + ".top2bottom_dw_conv.conv.weight"], a["pool"], bn=bn(pre + ".top2bottom_dw_conv.bn"), mode=1, ld=C)
                plan.add(sd[pre + ".top2bottom_pw_conv.conv.weight"], a["pw"].weight)
                plan.add(sd[pre + ".top2bottom_pw_conv.conv.weight"], None, bn=bn(pre + ".top2bottom_dw_conv.bn"), bias_out=a["pw"].bias,
                         bias_mode=3)
                plan.add(sd[pre + ".bottom_dw_conv.conv.weight"], a["dww"], bn=bn(pre + ".bottom_dw_conv.bn"), mode=1, ld=self.D,
                         bias_out=a["dwb"], bias_mode=1)
            return plan.finalize()
        except (AssertionError, KeyError):
            return None


    @property
    def logit_scale_exp(self):
        if self._ls_event is not None:
            self._ls_event.synchronize()
            self._ls_val = float(self._ls_host[0])
            self._ls_event = None
        return self._ls_val

    # ------------------------------------------------------------------ workspace
    def _workspace(self, Bi, Bt, inference=False):
        key = (Bi, Bt)
        w = self._ws.get(key)
        if w is not None and inference and w.get("held"):
            # a training forward's conv maps live in this workspace until its backward has run (train.TrainStep): an
            # inference call of the same shape in between gets a workspace of its own
            key = (Bi, Bt, "inference")
            w = self._ws.get(key)
        if w is not None:
            self._ws[key] = self._ws.pop(key)                  # most recently used last
            return w
        if not torch.cuda.is_current_stream_capturing():
            # keep at most two eager workspaces (e.g. the steady batch and a ragged last batch): a third shape evicts the
            # least recently used one instead of growing without bound (graph captures own theirs)
            eager = [k for k in self._ws if isinstance(k, tuple) and len(k) in (2, 3) and k[0] != "graph"
                     and not self._ws[k].get("pinned") and not self._ws[k].get("held")]
            for k in eager[:-1] if len(eager) >= 2 else []:
                del self._ws[k]
        dev, D, E = self.dev, self.D, self.E
        bf, f32 = torch.bfloat16, torch.float32
        Mv, Mt = Bi * self.Lv, Bt * self.Lt
        # buffers hold the text rows rounded up to whole 256-row tiles (the padded total of a packed batch can reach that);
        # w["M"] / w["Mt"] are what the current call runs over (set per call: _text_sizes / _text_dynamic / _text_unpacked)
        Mt_cap = -(-Mt // 256) * 256
        M_rows, M = Mv + Mt_cap, Mv + Mt

        def buf(*shape, dtype=bf):
            if len(shape) == 2 and dtype == bf:        # 64 elements of finite slack behind every bf16 matrix
                n = shape[0] * shape[1]
                flat = torch.zeros(n + 64, dtype=dtype, device=dev)
                return flat[:n].view(shape)
            return torch.empty(shape, dtype=dtype, device=dev)

        w = dict(Mv=Mv, Mt=Mt, M=M, Mt_cap=Mt_cap)
        M = M_rows                                           # (allocation size of everything below)
        w["X"] = buf(M, D, dtype=f32)
        w["LNO"], w["QKV"], w["AO"], w["HID"] = buf(M, D), buf(M, 3 * D), buf(M, D), buf(M, 4 * D)
        # LayerNorm fold (_blocks_fold): per-row centre (the row's mean at the previous LayerNorm point), (rstd, mean * rstd) for
        # the consuming projection, the producing GEMM's per-64-column partial sums; which row segments hold produced operands
        w["CEN"], w["RST"], w["PART"] = buf(M, dtype=f32), buf(M, 2, dtype=f32), buf(M, D // 64, 2, dtype=f32)
        w["fold_pending"] = {"v": False, "t": False}
        if self.fp8:                                         # e4m3 LayerNorm output + its per-token scales; e4m3 MLP hidden
            w["LNQ"] = torch.zeros(M * D + 256, dtype=torch.uint8, device=dev)[:M * D].view(M, D)
            w["RS"] = torch.empty(M, dtype=f32, device=dev)
            w["HIDQ"] = torch.zeros(M * 4 * D + 256, dtype=torch.uint8, device=dev)[:M * 4 * D].view(M, 4 * D)
            w["ONES"] = torch.ones(M, dtype=f32, device=dev)
        if Bi and self.patch:
            w["PATCH"] = buf(Bi * self.g * self.g, self.patch_k)     # bf16 patch matrix of the patch-conv stem (msclip_patchify)
            w["Ts"] = None
            w["hv"] = buf(Bi, D)
            w["fv_raw"], w["fv"] = buf(Bi, E, dtype=f32), buf(Bi, E, dtype=f32)
        elif Bi:
            w["XA"] = buf(Mv, D, dtype=f32)
            h1 = self.h1
            w["P0"] = buf(Bi * h1 * h1, D // 16)            # S1 (conv1's map) only exists on the unfused path: _s1()
            w["stem"] = [buf(Bi * s.h_out * s.w_out, s.cout) for s in self.stem_specs]
            w["par"] = [w["P0"]]
            w["par_tmp"] = [None]
            for j in range(1, 5):
                c1, c2, cr, c3 = self.par_specs[j]
                w["par_tmp"].append((buf(Bi * c1.h_out * c1.w_out, c1.cout), buf(Bi * c2.h_out * c2.w_out, c2.cout),
                                     buf(Bi * cr.h_out * cr.w_out, cr.cout)))
                w["par"].append(buf(Bi * c3.h_out * c3.w_out, c3.cout))
            w["pool"] = [buf(Bi * self.g * self.g, a["C"]) for a in self.adapters]
            w["T"] = buf(Bi * self.g * self.g, D, dtype=f32)
            w["Ts"] = None                                  # per-adapter buffers of the side-stream schedule (allocated on first use)
            w["hv"] = buf(Bi, D)
            w["fv_raw"], w["fv"] = buf(Bi, E, dtype=f32), buf(Bi, E, dtype=f32)
        if Bt:
            w["eot"] = torch.empty(Bt, dtype=torch.int32, device=dev)
            # device-side row counts of a packed batch (dynamic_rows): lengths, prefix sums and msclip_text_lengths' dims block
            w["len_d"] = torch.empty(Bt, dtype=torch.int32, device=dev)
            w["cu_d"] = torch.empty(Bt + 2, dtype=torch.int32, device=dev)
            w["dims"] = torch.zeros(8, dtype=torch.int32, device=dev)
            w["dyn"] = False
            w["packed"] = False                             # packed captions: w["cap"] = the staged batch (Captions) of this call
            w["ht"] = buf(Bt, D)
            w["ft_raw"], w["ft"] = buf(Bt, E, dtype=f32), buf(Bt, E, dtype=f32)
        # compact matrices of the rows that are still read after the last block's attention (cls / EOT rows, _last_block_tail)
        nc = Bi + Bt
        w["XC"] = buf(nc, D, dtype=f32)
        w["AOC"], w["LNC"], w["HIDC"], w["QC"] = buf(nc, D), buf(nc, D), buf(nc, 4 * D), buf(nc, D)
        if Bi:
            w["fvb"] = buf(Bi, E)                            # bf16 unit features: gather payload / logits operand
        if Bt:
            w["ftb"] = buf(Bt, E)
        self._ws[key] = w
        return w

    # ------------------------------------------------------------------ pieces
    def _conv(self, x, spec, out, B, act=hip.ACT_NONE, resid=None):
        rk = hip.RESID_BF16 if resid is not None else hip.RESID_NONE
        M = B * spec.h_out * spec.w_out
        if spec.kh == 1 and spec.kw == 1 and spec.stride == 1 and spec.pad == 0:
            # pointwise conv on NHWC == dense GEMM with ldx = Cin.  K is padded to 64: the tail chunks of a row read
            # the next row's (finite) activations against zero weights; activation buffers carry 64 elements of slack.
            return hip.gemm(x, spec.weight, out, M=M, N=spec.cout, bias=spec.bias, act=act, resid=resid,
                            resid_kind=rk, ldx=spec.cin)
        return hip.gemm(x, spec.weight, out, M=M, N=spec.cout, bias=spec.bias, act=act, resid=resid, resid_kind=rk,
                        conv=spec.geometry(), ktab=spec.ktab)

    @staticmethod
    def _fusable_3x3s2(spec):
        """3x3 / stride 2 / pad 1 over 48 channels: what the fused front kernels (csrc/front.hip) consume."""
        return (spec.kh, spec.kw, spec.stride, spec.pad, spec.cin) == (3, 3, 2, 1, 48) and spec.cout in (48, 96)

    def _tap_nhwc(self, taps, name, t, Bi, h, c):
        if taps is not None:
            taps[name] = t[:Bi * h * h].view(Bi, h, h, c).permute(0, 3, 1, 2).float()        # reference layout: NCHW

    def _tap_tokens(self, taps, name, t, n, L):
        if taps is not None:
            taps[name] = t[:n * L].view(n, L, -1).float().clone()                            # batch-first [B, L, C]

    def _vision_front(self, img, w, Bi, taps=None, keep_pre=None, convs_done=False):
        """Stem + tokenisation (M.py:2416-2426) and stage 0 of the parallel branch (M.py:2436).  convs_done: the stem's maps
        are already in the workspace (the training step's train-mode BatchNorm chain), only the tokenisation runs."""
        if self.patch:                                       # M.py:2657: conv1 as patchify + one dense GEMM (scatter + positional add fused)
            hip.patchify(img, w["PATCH"], Bi, self.S, self.S // self.g, self.patch_k)
            return self._tokenise(w["PATCH"], w, Bi, taps, keep_pre, weight=self.w_patch)
        if convs_done:
            return self._tokenise(w["stem"][-1], w, Bi, taps, keep_pre)
        first = self.stem_specs[0]
        fused = (self.dual_w.shape[1] == 96 and self._fusable_3x3s2(first) and not (self.opt.front_unfused or self.force_unfused)
                 and img.numel() * img.element_size() < 2 ** 31
                 and Bi * self.h1 * self.h1 * (self.D // 16) * 2 < 2 ** 31)      # the kernel's own 32-bit offset limits
        if fused:
            # conv1 + parallel stage 0 + stem stage 0 in one pass: conv1's 48-channel map never reaches HBM
            hip.stem_dual_conv3x3s2(img, self.dual_w, self.dual_b, w["P0"], first.weight, first.bias, w["stem"][0])
            if self.opt.branch_early and self._multi_stream_ok():
                # the parallel branch reads P0 only: it may start here, beside the rest of the stem, instead of behind the whole
                # front (same-box alternating, three pairs: 10.349 / 10.361 / 10.364 -> 10.328 / 10.345 / 10.339 ms per C2 step)
                w["p0_ready"] = self._record(torch.cuda.current_stream(self.dev))
            x, rest = w["stem"][0], list(zip(self.stem_specs, w["stem"]))[1:]
        else:
            hip.stem_conv_dual(img, self.dual_w, self.dual_b, self._s1(w, Bi), w["P0"])
            x, rest = w["S1"], list(zip(self.stem_specs, w["stem"]))
        for spec, out in rest:
            self._conv(x, spec, out, Bi, act=hip.ACT_RELU)
            x = out
        if taps is not None:
            if not fused:
                self._tap_nhwc(taps, "stem_conv1", w["S1"], Bi, self.h1, self.D // 16)
            for i, (spec, out) in enumerate(zip(self.stem_specs, w["stem"])):
                self._tap_nhwc(taps, f"stem_stage{i}", out, Bi, spec.h_out, spec.cout)
            self._tap_nhwc(taps, "parallel0", w["P0"], Bi, self.h1, self.D // 16)
        self._tokenise(x, w, Bi, taps, keep_pre)

    def _tokenise(self, x, w, Bi, taps=None, keep_pre=None, weight=None):
        g2 = self.g * self.g
        # last_conv (1x1, no BN/ReLU) -- or the patch conv over the patch matrix -- fused with "+ positional_embedding" and the
        # scatter to token rows b*L + 1 + p
        hip.gemm(x, self.w_last if weight is None else weight, w["X"], M=Bi * g2, resid=self.vpos, resid_kind=hip.RESID_TABLE,
                 rpg=g2, radd=1, roff=1)
        hip.fill_cls(self.cls, self.vpos, w["X"], Bi, self.Lv)
        xv = w["X"][:w["Mv"]]
        if keep_pre is not None:
            keep_pre.append(xv.clone())                     # tokens before ln_pre (the training step's backward needs them)
        hip.layernorm(xv, self.ln_pre.g, self.ln_pre.b, xv, w["Mv"])
        self._tap_tokens(taps, "tokens_ln_pre", xv, Bi, self.Lv)

    def _parallel_stage(self, j, w, Bi):
        if j == 0:
            return
        c1, c2, cr, c3 = self.par_specs[j]
        t1, t2, tr = w["par_tmp"][j]
        src = w["par"][j - 1]
        lead = ((c1.kh, c1.kw, c1.stride, c1.pad, c1.cin, c1.cout) == (1, 1, 1, 0, 48, 48) and self._fusable_3x3s2(c2)
                and not (self.opt.front_unfused or self.force_unfused) and src.numel() * 2 < 2 ** 31)
        if (lead and c2.cout == 48 and (cr.kh, cr.kw, cr.stride, cr.pad, cr.cin, cr.cout) == (1, 1, 2, 0, 48, 96)
                and (c3.kh, c3.kw, c3.stride, c3.pad, c3.cin, c3.cout) == (1, 1, 1, 0, 48, 96)
                and (cr.h_out, cr.w_out) == (c2.h_out, c2.w_out) and not self.opt.block_unfused):
            # the whole stride-2 bottleneck in one launch: conv1's map, conv2's output and the shortcut stay on chip
            hip.convresblock48_s2(src, c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, cr.weight, self.par_b3r[j],
                                  w["par"][j], Bi, c1.h_in, c1.w_in)
            return
        if lead:
            # conv1 -> conv2 of the bottleneck without its 112 x 112 intermediate map
            hip.conv1x1_conv3x3s2(src, c1.weight, c1.bias, c2.weight, c2.bias, t2, Bi, c1.h_in, c1.w_in)
        else:
            self._conv(src, c1, t1, Bi, act=hip.ACT_RELU)
            self._conv(t1, c2, t2, Bi, act=hip.ACT_RELU)
        self._conv(src, cr, tr, Bi)
        self._conv(t2, c3, w["par"][j], Bi, act=hip.ACT_RELU, resid=tr)

    def _adapter_top(self, j, w, Bi, out):
        """Top-down half of Lateral_Adapter (M.py:1752-1762): depthwise k = s pooling conv + BN + pointwise conv -> out."""
        a = self.adapters[j]
        hw = self.par_hw[j]
        hip.dwpool(w["par"][j], a["pool"], w["pool"][j], Bi, hw, hw, a["C"], a["k"])
        pw = a["pw"]
        hip.gemm(w["pool"][j], pw.weight, out, M=Bi * self.g * self.g, N=pw.cout, bias=pw.bias, ldx=pw.cin)

    def _adapter(self, j, w, Bi, t=None, ln1=None):
        """Lateral_Adapter (M.py:1752-1778): X[:Mv] -> XA.  `t`: the top-down half if it was already computed.  `ln1`: the
        LayerNorm of the block behind the adapter, applied to each row while it is in registers (-> LNO, CEN, RST; M.py:1027)."""
        a = self.adapters[j]
        if t is None:
            t = w["T"]
            self._adapter_top(j, w, Bi, t)
        if ln1 is not None:
            Mv = w["Mv"]
            return hip.adapter_combine_ln_stats(w["X"][:Mv], t, a["dww"], a["dwb"], a["ln"].g, a["ln"].b, w["XA"], ln1.g, ln1.b,
                                                w["LNO"][:Mv], w["CEN"][:Mv], w["RST"][:Mv], Bi, self.Lv, self.g, self.usecls)
        hip.adapter_combine_ln(w["X"][:w["Mv"]], t, a["dww"], a["dwb"], a["ln"].g, a["ln"].b, w["XA"], Bi,
                               self.Lv, self.g, self.usecls)

    def _conv_branch_on_side_stream(self, w, Bi):
        """The parallel convolutional branch and the adapters' top-down halves depend on the image only (M.py:2436,
        2128-2159), not on the token stream: issue all of them on a side HIP stream right after the front pass and let
        each adapter wait for its own event.  They are HBM-bound, small-LDS kernels; the dispatcher places their
        workgroups on CUs the persistent 160-KiB-LDS GEMM workgroups have left (launch tails, tile-count remainders)
        and beside the LayerNorm / attention launches.  Same kernels, same data, same results."""
        if w["Ts"] is None:
            n = Bi * self.g * self.g
            w["Ts"] = [torch.empty(n, self.D, dtype=torch.float32, device=self.dev) for _ in self.adapters]
        cur = torch.cuda.current_stream(self.dev)
        side = self.conv_stream()
        ready = w.pop("p0_ready", None)
        if ready is None:
            ready = self._record(cur)                       # the front pass (parallel stage 0's map) is queued
        self._wait(side, ready)
        events = []
        with torch.cuda.stream(side):
            for j in range(len(self.adapters)):
                self._parallel_stage(j, w, Bi)
                self._adapter_top(j, w, Bi, w["Ts"][j])
                events.append(self._record(side))
        return events

    def conv_stream(self):
        """The side stream of the image-only conv branch: an ordinary stream, or -- EngineOptions.side_cu_mask > 0 -- one confined
        to that many CUs (hipExtStreamCreateWithCUMask), so that the branch's HBM-bound workgroups stop displacing the
        persistent GEMM workgroups of the main stream."""
        n = self.opt.side_cu_mask
        if n <= 0 and self.opt.side_priority == "low":
            key = (self.dev, "low")
            if key not in _MASKED:
                _MASKED[key] = hip.priority_stream(self.dev, urgent=False)
            return _MASKED[key]
        if n <= 0:
            return C.side_stream(self.dev)
        key = (self.dev, n)
        if key not in _MASKED:
            _MASKED[key] = hip.cu_masked_stream(self.dev, n)
        return _MASKED[key]

    # ---- cross-stream edges: a torch event for the eager pass, and the same edge in the launch plan when one is recording
    def _record(self, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        ev._plan_id = self._rec.record_event(stream) if self._rec is not None else -1
        return ev

    def _wait(self, stream, ev):
        stream.wait_event(ev)
        if self._rec is not None:
            self._rec.wait_event(stream, ev._plan_id)

    def lateral_on_last(self):
        return (self.n_layers - 1) in self.lateral

    def _s1(self, w, Bi):
        if "S1" not in w:
            n = Bi * self.h1 * self.h1 * (self.D // 16)
            w["S1"] = torch.zeros(n + 64, dtype=torch.bfloat16, device=self.dev)[:n].view(-1, self.D // 16)
        return w["S1"]

    # ------------------------------------------------------------------ packed (pad-free) captions
    def text_pack_enabled(self):
        """Captions run PACKED by default: under the causal mask (M.py:2965-2971) no row behind a caption's EOT position can
        reach the EOT row encode_text returns (M.py:3057-3060), in any block, so caption b owns n_b = argmax + 1 rows of the
        token matrix instead of 77.  Identical features / logits / loss / gradients; the token matrix's row count becomes
        data-dependent: it stays on the device (dynamic_rows: every launch reads it there) or -- small batches, the training step,
        EngineOptions.dynamic_rows = False -- is read by the host once per call, BEFORE anything of the call is queued
        (stage_captions hides that read).  EngineOptions.text_pack = False: every caption computes all context_length rows."""
        return self.opt.text_pack and self.Lt <= 96

    def _multi_stream_ok(self):
        """The eager launch loop uses side streams unless the caller is capturing a hipGraph (a capture of a PLAN replay has them:
        msclip_plan_run forks and joins its side streams with the plan's own events)."""
        return not torch.cuda.is_current_stream_capturing()

    def dynamic_rows(self, Bi, Bt):
        """Packed captions with the row count kept ON THE DEVICE (msclip_text_lengths' dims, msclip_gemm_desc.M_dev): every launch
        over the text rows is sized for the upper bound -- all context_length rows, rounded up to whole 256-row tiles -- and reads
        the batch's padded total itself.  No host read, so a call never waits for the GPU, a hipGraph capture records the packed
        step, and one recorded launch table (msclip_plan_*) serves every batch.  Needs the LayerNorm-fold form of the layer loop
        (whole tiles that never straddle the image / text boundary); otherwise the round-5 path (host reads the total) runs."""
        if not (Bt and self.opt.dynamic_rows and self.text_pack_enabled()) or self.fused_qkv_attn_enabled() or self._calib is not None:
            return False
        if self.fp8_qkv or not self.opt.ln_fold or self.D % 256:
            return False
        cap = -(-(Bt * self.Lt) // 256) * 256
        return cap >= 256 * 16 and (not Bi or ((Bi * self.Lv) % 256 == 0 and Bi * self.Lv >= 256 * 16))

    def stage_captions(self, tok):
        """-> Captions: queue the length / prefix-sum kernels of a token batch and the 8-byte read-back on the CURRENT stream
        (ordered behind whatever produced `tok` there).  run() / forward_* / encode_text / TrainStep.forward take the result in
        place of the token tensor; called on a tensor they stage it themselves and wait for the read-back at once."""
        tok = self._check_tok(tok)
        Bt = tok.shape[0]
        with torch.cuda.device(self.dev):
            length = torch.empty(Bt, dtype=torch.int32, device=self.dev)
            cu = torch.empty(Bt + 2, dtype=torch.int32, device=self.dev)
            eot = torch.empty(Bt, dtype=torch.int32, device=self.dev)
            hip.text_lengths(tok, length, cu, eot, row_base=0)
            # pinned read-back slots come from a free list and return to it when their batch's totals have been read (round 5 used
            # a 64-slot ring without an overwrite guard: a deep prefetch queue could have read another batch's row total)
            pool = self.__dict__.setdefault("_cap_pool", [])
            host = pool.pop() if pool else torch.zeros(2, dtype=torch.int32).pin_memory()
            host.copy_(cu[Bt:Bt + 2], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
        return Captions(tok, length, cu, eot, host, ev, release=pool.append)

    def _text_sizes(self, cap, w, Bt):
        """The host read of a staged batch -> this call's row counts (w["Mt"], w["M"], w["pad"])."""
        total, lmax = cap.totals()
        # the staged batch's device tensors were allocated on the STAGING stream (an input pipeline's prefetch stream): tell the
        # caching allocator that the stream consuming them is another one, or a block freed when the next batch replaces this one
        # could be handed to the next stage_captions while this step's kernels still read it
        cur = torch.cuda.current_stream(self.dev)
        for t in (cap.tok, cap.len, cap.cu, cap.eot):
            t.record_stream(cur)
        padded = -(-total // 256) * 256              # whole 256-row GEMM tiles (the LayerNorm fold's modality split never straddles one)
        if padded > Bt * self.Lt or padded < 256 * 16:
            padded = total
        w.update(packed=True, dyn=False, cap=cap, cu=cap.cu, len=cap.len, Mt_live=total, Lmax=lmax, pad=padded - total, Mt=padded,
                 M=w["Mv"] + padded, mdev_t=None, mdev_all=None)

    def _text_unpacked(self, w, Bt):
        w.update(packed=False, dyn=False, cap=None, Mt_live=Bt * self.Lt, Lmax=self.Lt, pad=0, Mt=Bt * self.Lt,
                 M=w["Mv"] + Bt * self.Lt, mdev_t=None, mdev_all=None)

    def _text_dynamic(self, w, Bt, lmax=None):
        """Packed captions whose row count stays on the device (dynamic_rows): the call runs over the upper bound w["Mt_cap"] and
        every launch over the text rows gets a device counter (mdev_t: padded text rows, mdev_all: rows of the whole matrix) or
        the dims block.  The lengths themselves are computed by _text_front, as the first launches of the text front.  lmax: the
        longest caption if the host happens to know it (a batch staged ahead whose read-back has landed): picks the attention
        kernel's tile count; None = context_length."""
        cap = w["Mt_cap"]
        w.update(packed=True, dyn=True, cap=None, cu=w["cu_d"], len=w["len_d"], Mt_live=None, Lmax=lmax or self.Lt, pad=255, Mt=cap,
                 M=w["Mv"] + cap, mdev_t=w["dims"][2:3], mdev_all=w["dims"][3:4])

    def _text_lengths_dynamic(self, tok, w):
        """The two tiny launches behind every device-side row count: lengths, prefix sums, EOT rows (absolute: row_base = Mv) and the
        dims block.  Queued on the MAIN stream before the text stream forks off: a 1-workgroup scan that has to find a CU beside
        the persistent front kernel measured 334 us instead of 6 (profiles/r06_forward_timeline.txt of the first version)."""
        hip.text_lengths(tok, w["len_d"], w["cu_d"], w["eot"], row_base=w["Mv"], dims=w["dims"], pad_to=256, cap_rows=w["Mt_cap"])

    @staticmethod
    def _md(w, r0, r1):
        """The device-side row counter of a launch over rows [r0, r1) of the token matrix (None: the rows are a host-side constant)."""
        if not w.get("dyn") or r1 != w["M"]:
            return None
        return w["mdev_all"] if r0 == 0 else w["mdev_t"]

    def _live_text_rows(self, w):
        """Live text rows of this call on the HOST (taps / calibration / accounting only: a dynamic-rows call reads it back)."""
        if w.get("Mt_live") is None:
            w["Mt_live"] = int(w["dims"][0].item())
        return w["Mt_live"]

    def _text_front(self, tok, w, Bt):
        if w.get("dyn"):
            # (lengths / prefix sums / EOT rows / dims block: _text_lengths_dynamic, queued by the caller in front of the stream fork)
            return hip.embed_tokens_packed(tok, self.emb, self.tpos, w["X"], w["cu"], w["Mv"], w["Mt"], rows_dev=w["mdev_t"])
        if w.get("packed"):
            torch.add(w["cap"].eot, w["Mv"], out=w["eot"])           # EOT rows of the packed segment -> rows of the token matrix
            return hip.embed_tokens_packed(tok, self.emb, self.tpos, w["X"], w["cu"], w["Mv"], w["Mt"])
        hip.embed_tokens(tok, self.emb, self.tpos, w["X"], w["eot"], w["Mv"])

    def _attention_text(self, w, QKV, AO, Bt):
        Mv, M = w["Mv"], w["M"]
        if w.get("packed"):
            # (always the context-length instantiation: on packed captions the 3-tile kernel measures 34 us where the 2-tile one
            #  a longest caption <= 64 would select takes 47 us, profiles/r06_attention_wpb_occupancy_probe.txt)
            return hip.attention_varlen(QKV[Mv:M], AO[Mv:M], w["cu"], Bt, max(w["Lmax"], min(self.Lt, 96)), self.heads, True,
                                        pad_rows=w["pad"], dims=w["dims"] if w.get("dyn") else None)
        hip.attention(QKV[Mv:M], AO[Mv:M], Bt, self.Lt, self.heads, True)

    def _tap_text(self, taps, name, t, w, Bt):
        """Text-block tap in the reference's [B, L, C] layout; packed captions: live rows in place, the others zero, and
        taps["text_lengths"] says which are which."""
        if taps is None:
            return
        if not w.get("packed"):
            return self._tap_tokens(taps, name, t, Bt, self.Lt)
        lens = w["len"].long()
        cu = w["cu"][:Bt].long()
        pos = torch.arange(self.Lt, device=self.dev)[None, :]
        live = pos < lens[:, None]
        rows = (cu[:, None] + pos).clamp_(max=self._live_text_rows(w) - 1)
        out = t[:w["Mt"]].float()[rows.reshape(-1)].view(Bt, self.Lt, -1)
        taps[name] = out * live[:, :, None]
        taps["text_lengths"] = w["len"].clone()

    def _last_block_attention(self, w, Bi, Bt, groups):
        """The last block's attention when only the class / EOT rows are read afterwards: keys and values are still projected
        for every token (the k | v two thirds of in_proj), the query only for the Bi + Bt live rows, and one query per
        sample attends (msclip_attention_lastq) straight into the compact matrix AOC."""
        D, Mv = self.D, w["Mv"]
        LNO, QKV, LNC, QC, AOC = w["LNO"], w["QKV"], w["LNC"], w["QC"], w["AOC"]
        for r0, r1, bw in groups:
            hip.gemm(LNO[r0:r1], bw.wqkv[D:], QKV[r0:r1, D:], bias=bw.bqkv[D:], mdev=self._md(w, r0, r1))
        if Bi:
            hip.gather_rows(LNO, LNC[:Bi], Bi, row_mul=self.Lv)
        if Bt:
            hip.gather_rows(LNO, LNC[Bi:], Bt, row_idx=w["eot"])
        n = Bi + Bt
        cgroups = [(0, n, groups[0][2])] if len(groups) == 1 else [(0, Bi, groups[0][2]), (Bi, n, groups[1][2])]
        for r0, r1, bw in cgroups:
            if r1 > r0:
                hip.gemm(LNC[r0:r1], bw.wqkv[:D], QC[r0:r1], bias=bw.bqkv[:D])
        if Bi:
            hip.attention_lastq(QC[:Bi], QKV, AOC[:Bi], Bi, self.Lv, self.heads)
        if Bt and w.get("packed"):
            hip.attention_lastq_varlen(QC[Bi:], QKV, AOC[Bi:], Bt, w["Lmax"], self.heads, w["cu"], row_base=Mv)
        elif Bt:
            hip.attention_lastq(QC[Bi:], QKV, AOC[Bi:], Bt, self.Lt, self.heads, last_row=w["eot"], row_base=Mv)

    def _last_block_tail(self, w, Bi, Bt, vb, tb, attended=False):
        """After the last block's attention only x[:, 0, :] of every image (M.py:2685) and the EOT row of every caption
        (M.py:3057-3060) are read again, and out_proj / ln_2 / c_fc / c_proj are row-wise: they run on those Bi + Bt rows,
        moved to the compact matrices XC (fp32 stream) / AOC (attention output), instead of on all tokens.  Same results
        for encode_image / encode_text / forward / the loss; the other rows of X keep their pre-out_proj values."""
        XC, AOC, LNC, HIDC = w["XC"], w["AOC"], w["LNC"], w["HIDC"]
        X, AO = w["X"], w["AO"]
        if Bi:
            hip.gather_rows(X, XC[:Bi], Bi, row_mul=self.Lv)
            if not attended:
                hip.gather_rows(AO, AOC[:Bi], Bi, row_mul=self.Lv)
        if Bt:
            hip.gather_rows(X, XC[Bi:], Bt, row_idx=w["eot"])
            if not attended:
                hip.gather_rows(AO, AOC[Bi:], Bt, row_idx=w["eot"])
        n = Bi + Bt
        segs = ([(0, Bi, vb)] if Bi else []) + ([(Bi, n, tb)] if Bt else [])
        groups = [(0, n, vb["w"])] if len(segs) == 2 and vb["w"] is tb["w"] else [(r0, r1, b["w"]) for r0, r1, b in segs]
        for r0, r1, bw in groups:
            hip.gemm(AOC[r0:r1], bw.wo, XC[r0:r1], bias=bw.bo, resid=XC[r0:r1], resid_kind=hip.RESID_F32)
        if len(segs) == 2:
            hip.layernorm_split(XC[:n], vb["ln2"].g, vb["ln2"].b, tb["ln2"].g, tb["ln2"].b, Bi, LNC[:n], n)
        else:
            for r0, r1, b in segs:
                hip.layernorm(XC[r0:r1], b["ln2"].g, b["ln2"].b, LNC[r0:r1], r1 - r0)
        for r0, r1, bw in groups:
            hip.gemm(LNC[r0:r1], bw.wfc, HIDC[r0:r1], bias=bw.bfc, act=hip.ACT_QUICKGELU)
            hip.gemm(HIDC[r0:r1], bw.wpr, XC[r0:r1], bias=bw.bpr, resid=XC[r0:r1], resid_kind=hip.RESID_F32)

    def _mlp_f8(self, w, r0, r1, bw, fold_out=None):
        """c_fc + QuickGELU + c_proj of the rows [r0, r1) under PRECISION fp8 (recipe: DESIGN.md s9, emulated by
        oracle/fp8_recipe.py).  c_fc (e4m3 LayerNorm output x e4m3 weight, per-token x per-channel scales) writes the hidden
        matrix as e4m3 with ONE static scale per layer -- its row maximum spans 16-32 column tiles, so a per-token scale cannot
        come out of a tile's epilogue -- which makes it the fp8 operand of c_proj without another pass.  The scale comes from
        calibrate_fp8() (explicit; both modalities of a shared layer, all ranks): an uncalibrated layer, and row counts that are
        not whole 256-row tiles, keep a bf16 hidden matrix and a bf16 c_proj."""
        X, HID, LNQ, RS = w["X"], w["HID"], w["LNQ"], w["RS"]
        md = self._md(w, r0, r1)
        if bw.hid_scale is None or (r1 - r0) % 256:
            hip.gemm_f8(LNQ[r0:r1], bw.wfc_q, HID[r0:r1], RS[r0:r1], bw.wfc_s, bias=bw.bfc, act=hip.ACT_QUICKGELU, mdev=md)
            if self._calib is not None:                    # calibrate_fp8(): max |hidden| of this launch, kept on the device
                # ... over the rows that hold real tokens: the tile padding behind the last caption (zero embeddings run through
                # the blocks) must not set a scale
                live = min(r1, w["Mv"] + self._live_text_rows(w)) if (w.get("packed") and r1 > w["Mv"]) else r1
                self._calib.setdefault(id(bw), (bw, []))[1].append(HID[r0:live].abs().amax())
            hip.gemm(HID[r0:r1], bw.wpr, X[r0:r1], bias=bw.bpr, resid=X[r0:r1], resid_kind=hip.RESID_F32, mdev=md)
            return
        HQ = w["HIDQ"]
        hip.gemm_f8(LNQ[r0:r1], bw.wfc_q, HQ[r0:r1], RS[r0:r1], bw.wfc_s, bias=bw.bfc, act=hip.ACT_QUICKGELU,
                    out_scale=1.0 / bw.hid_scale, mdev=md)
        hip.gemm_f8(HQ[r0:r1], bw.wpr_q, X[r0:r1], w["ONES"][r0:r1], bw.wpr_cs, bias=bw.bpr, resid=X[r0:r1], resid_kind=hip.RESID_F32,
                    fold_out=fold_out, mdev=md)

    # ------------------------------------------------------------------ fp8 calibration (PRECISION fp8 / fp8-qkv)
    HID_HEADROOM = 1.25                                   # static hidden scale = HID_HEADROOM * calibrated max |hidden| / 448

    def _fp8_blocks(self):
        """{(layer, tower): _BlockW} of every packed block (shared layers appear under both towers with the same object)."""
        out = {}
        for i in range(self.n_layers):
            for tower, blks in (("v", self.vblk), ("t", self.tblk)):
                if blks[i] is not None:
                    out[(i, tower)] = blks[i]["w"]
        return out

    def fp8_calibrated(self):
        """True once calibrate_fp8 has run (or scales were loaded).  Blocks whose MLP never runs on the token matrix -- the last
        block's live-row tail -- have no scale and need none."""
        return not self.fp8 or bool(self._fp8_saved)

    def calibrate_fp8(self, img, tok):
        """Fix the static e4m3 scale of every layer's MLP hidden matrix from ONE calibration batch that has both modalities
        (a shared layer's hidden activations differ between image and text rows; the scale covers both): the batch runs with
        bf16 hidden matrices, each layer's max |hidden| stays on the device, the maxima are all-reduced (MAX) over the ranks
        of a process group -- every rank ends with the same scales -- and read by the host once.  Scales survive re-packs
        (optimizer steps, load_state_dict): call again to re-calibrate; fp8_state() / load_fp8_state() carry them in a checkpoint."""
        if not self.fp8:
            return {}
        if img is None or tok is None:
            raise ValueError("calibrate_fp8 needs images AND captions: the shared layers' hidden scale must cover both modalities")
        self.refresh()                                       # weights changed outside TrainStep.step: re-pack BEFORE clearing (the
        saved, self._fp8_saved = self._fp8_saved, {}         # re-pack would otherwise restore the old scales into fresh copies)
        blocks = self._fp8_blocks()
        for bw in blocks.values():
            bw.hid_scale, bw.wpr_cs = None, None
        self._calib = {}
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        failure, objs, amax = None, [], None
        try:
            self.run(img, tok)
            if not self._calib:
                raise RuntimeError("calibrate_fp8: no layer recorded a hidden-matrix maximum (did the batch reach the fp8 MLP?)")
            objs = [bw for bw, _ in self._calib.values()]
            amax = torch.stack([torch.stack(vals).amax() for _, vals in self._calib.values()])
        except Exception as exc:                             # (rank-local: the other ranks are about to enter a collective)
            failure = exc
        finally:
            self._calib = None
        if multi:
            # agree on success BEFORE the collective over the maxima: a rank that failed locally must not leave the others
            # hanging in the all-reduce -- every rank learns that some rank failed and raises
            flag = torch.tensor([0.0 if failure is None else 1.0], device=self.dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if failure is None and flag.item() > 0:
                failure = RuntimeError("calibrate_fp8 failed on another rank")
        if failure is not None:
            if saved:                                        # a failed re-calibration keeps the previous scales
                self.load_fp8_state(saved)
            raise failure
        if multi:
            dist.all_reduce(amax, op=dist.ReduceOp.MAX)
        host = amax.cpu().tolist()                           # the one host read of the calibration
        for bw, a in zip(objs, host):
            self._set_hid_scale(bw, max(a, 1e-6) * self.HID_HEADROOM / 448.0)
        self._fp8_saved = self.fp8_state()
        return dict(self._fp8_saved)

    @staticmethod
    def _set_hid_scale(bw, s):
        bw.hid_scale = float(s)
        bw.wpr_cs = (bw.wpr_s * bw.hid_scale).contiguous()

    def fp8_state(self):
        """{"layer.tower": hidden scale} of the calibrated layers (plain floats: goes into a checkpoint next to the weights)."""
        return {f"{i}.{t}": bw.hid_scale for (i, t), bw in self._fp8_blocks().items() if bw.hid_scale is not None}

    def load_fp8_state(self, state):
        blocks = self._fp8_blocks()
        for key, s in state.items():
            i, t = key.split(".")
            if (int(i), t) in blocks:
                self._set_hid_scale(blocks[(int(i), t)], s)
        self._fp8_saved = self.fp8_state()

    def _ln_f8(self, w, segs, which):
        """LayerNorm of the token rows straight to e4m3 + per-token scales (w["LNQ"], w["RS"]): one launch over both towers'
        rows (own gamma / beta per modality), or one per tower when only one runs."""
        X = w["X"]
        if len(segs) == 2:
            (r0, rs, vb), (_, r1, tb) = segs
            hip.layernorm_f8(X[r0:r1], vb[which].g, vb[which].b, tb[which].g, tb[which].b, rs - r0, w["LNQ"][r0:r1], w["RS"][r0:r1], r1 - r0,
                             mdev=self._md(w, r0, r1))
        else:
            for r0, r1, b in segs:
                hip.layernorm_f8(X[r0:r1], b[which].g, b[which].b, b[which].g, b[which].b, r1 - r0, w["LNQ"][r0:r1], w["RS"][r0:r1], r1 - r0,
                                 mdev=self._md(w, r0, r1))

    # ------------------------------------------------------------------ LayerNorm fold
    def _fold_eligible(self, w, Bi, Bt):
        """The fold runs on whole 256-row tiles of the ping-pong GEMM that never straddle the image / text boundary."""
        # PRECISION fp8 (c_fc / c_proj on the fp8 MFMA): ln_1 still folds -- in_proj is a bf16 consumer, the fp8 c_proj produces;
        # ln_2 feeds an e4m3 operand with per-token scales (a row maximum: needs the whole row) and keeps its pass.  fp8-qkv: no fold.
        if self.fp8_qkv or not self.opt.ln_fold or self.D % 256:
            return False
        rows = [n for n in ((w["Mv"] if Bi else 0), (w["M"] - w["Mv"] if Bt else 0)) if n]
        return bool(rows) and all(n % 256 == 0 and n >= 256 * 16 for n in rows)

    def _fold_weights(self, i, tower, which):
        """(W', csum, bias') of layer i's in_proj ("qkv") / c_fc ("fc") behind tower's ln_1 / ln_2 (M.py:1027-1028, 204-219):
        LN(x) W^T + b = rstd (x - mean) (gamma o W)^T + (b + W beta): W' = bf16(gamma o W) from the packed bf16 weight (the q rows
        carry their 64^-0.5), csum = row sums of the bf16 values (what the MFMA contracts), bias' in fp32."""
        key = (i, tower, which)
        if key not in self._foldw:
            blk = (self.vblk if tower == "v" else self.tblk)[i]
            bw, ln = blk["w"], blk["ln1" if which == "qkv" else "ln2"]
            W, b = (bw.wqkv, bw.bqkv) if which == "qkv" else (bw.wfc, bw.bfc)
            Wf = (W.float() * ln.g[None, :]).to(torch.bfloat16).contiguous()
            self._foldw[key] = (Wf, Wf.float().sum(dim=1).contiguous(), (b + W.float() @ ln.b).contiguous())
        return self._foldw[key]

    # ------------------------------------------------------------------ fused in_proj + attention (opt-in)
    def fused_qkv_attn_enabled(self):
        """MSCLIP_FUSED_QKV_ATTN=1: the layers whose attention tensors both towers share run in_proj + attention as ONE kernel
        (msclip_qkv_attention: q|k|v staged in LDS, never in HBM) -- what BASELINE.json's north_star names.  Off by default:
        its two-buffer 256 x 192 main loop is slower than the ping-pong GEMM by more than the attention launches cost
        (228 vs 204 us per layer at the packed C2 shapes)."""
        return self.opt.fused_qkv_attn and self.heads * 64 == self.D and self.Lv <= 96 and self.Lt <= 96

    def _fused_tables(self, w, Bi, Bt):
        """Row / tile tables of this call's token matrix: image samples of Lv rows, then the captions (packed: cu of the staged
        batch; full rows: Lt each).  Rebuilt per call (the captions change); two small launches."""
        Mv = w["Mv"]
        img = torch.arange(0, Mv, self.Lv, dtype=torch.int32, device=self.dev) if Bi else None
        if Bt and w.get("packed"):
            txt = w["cu"][:Bt] + Mv
            live = Mv + w["Mt_live"]
        elif Bt:
            txt = torch.arange(Mv, Mv + Bt * self.Lt, self.Lt, dtype=torch.int32, device=self.dev)
            live = Mv + Bt * self.Lt
        else:
            txt, live = None, Mv
        end = torch.full((1,), live, dtype=torch.int32, device=self.dev)
        cu = torch.cat([t for t in (img, txt, end) if t is not None]).contiguous()
        return hip.QkvAttnTables(cu, Bi + Bt, split_sample=Bi if (Bi and Bt) else 0, total_rows=live)

    def _fused_qkv_weights(self, i, tower, folded):
        """Head-major (weight, bias, csum) of layer i's in_proj for one tower's rows: gamma-folded (LNO holds x - centre) or
        plain (LNO holds the LayerNorm output; zero column sums)."""
        key = (i, tower, "qkv_hm", folded)
        if key not in self._foldw:
            if folded:
                W, c, b = self._fold_weights(i, tower, "qkv")
            else:
                bw = (self.vblk if tower == "v" else self.tblk)[i]["w"]
                W, b, c = bw.wqkv, bw.bqkv, torch.zeros(3 * self.D, dtype=torch.float32, device=self.dev)
            self._foldw[key] = hip.head_major_qkv(W, b, self.heads, c)
        return self._foldw[key]

    def _fused_qkv_attention(self, w, i, r0, r1, modes, Bi, Bt):
        """in_proj + attention of layer i over rows [r0, r1) = both towers (modes as in _fold_proj), one launch."""
        if w.get("fused_tabs") is None:
            w["fused_tabs"] = self._fused_tables(w, Bi, Bt)
            if w["M"] > w["fused_tabs"].rowseg.shape[0]:
                w["AO"][w["fused_tabs"].rowseg.shape[0]:w["M"]].zero_()     # tile padding behind the last caption: finite operand rows of out_proj
        Mv = w["Mv"]
        W1, b1, c1 = self._fused_qkv_weights(i, modes[0][0], modes[0][3])
        use_fold = any(m[3] for m in modes)
        fi = None
        if len(modes) == 2:
            W2, b2, c2 = self._fused_qkv_weights(i, modes[1][0], modes[1][3])
            fi = hip.FoldIn(w["RST"][r0:r1], c1, W2, b2, c2, modes[1][1] - r0)
        elif use_fold:
            fi = hip.FoldIn(w["RST"][r0:r1], c1)
        assert r0 == 0
        hip.qkv_attention(w["LNO"][r0:r1], W1, b1, w["AO"][r0:r1], w["fused_tabs"], self.heads,
                          causal_from_row=Mv if Bt else hip.INT_MAX, fold_in=fi, M=r1 - r0)

    def _fold_proj(self, w, i, which, r0, r1, modes, out, act):
        """One projection over rows [r0, r1) behind a LayerNorm.  modes = [(tower, row0, row1, folded)]: a folded segment's rows
        of LNO hold bf16 (x - centre) and take the gamma-folded weight; a plain segment's rows hold a LayerNorm output (RST rows
        (1, 0)) and take the packed weight as it is."""
        def seg_weights(tower, folded):
            if folded:
                return self._fold_weights(i, tower, which)
            bw = (self.vblk if tower == "v" else self.tblk)[i]["w"]
            W, b = (bw.wqkv, bw.bqkv) if which == "qkv" else (bw.wfc, bw.bfc)
            if "zcs" not in self._foldw:
                self._foldw["zcs"] = torch.zeros(4 * self.D, dtype=torch.float32, device=self.dev)
            return W, self._foldw["zcs"], b
        LNO, RST = w["LNO"], w["RST"]
        md = self._md(w, r0, r1)
        if not any(m[3] for m in modes):
            W, _, b = seg_weights(modes[0][0], False)
            return hip.gemm(LNO[r0:r1], W, out[r0:r1], bias=b, act=act, mdev=md)
        W1, c1, b1 = seg_weights(modes[0][0], modes[0][3])
        if len(modes) == 1:
            fi = hip.FoldIn(RST[r0:r1], c1)
        else:
            W2, c2, b2 = seg_weights(modes[1][0], modes[1][3])
            fi = hip.FoldIn(RST[r0:r1], c1, W2, b2, c2, modes[1][1] - r0)
        return hip.gemm(LNO[r0:r1], W1, out[r0:r1], bias=b1, act=act, fold_in=fi, mdev=md)

    def _blocks_fold(self, w, Bi, Bt, taps, conv_events, compact, layers):
        """The layer loop with the LayerNorms folded into the GEMMs around them (DESIGN.md "LayerNorm fold"): out_proj and c_proj
        write, beside the fp32 residual update, the bf16 operand of the projection that follows -- (x - centre[m]), the centre being
        the row's mean one LayerNorm earlier -- and per-row partial sums; a one-thread-per-row kernel turns those into (rstd, mean
        rstd); in_proj / c_fc apply them (and gamma / beta through their weights) to their accumulators.  The separate LayerNorm
        passes over the token matrix (a 200 MB read + 100 MB write each) remain only where the rows come from somewhere else:
        behind the stem / the embedding, behind a lateral adapter, and in the last block's live-row tail."""
        Mv, M, D = w["Mv"], w["M"], self.D
        X, LNO, QKV, AO, HID, CEN, RST, PART = (w[k] for k in ("X", "LNO", "QKV", "AO", "HID", "CEN", "RST", "PART"))
        pend = w["fold_pending"]
        n_last = self.n_layers - 1
        for i in (range(self.n_layers) if layers is None else layers):
            vb = self.vblk[i] if Bi else None
            tb = self.tblk[i] if Bt else None
            if vb is None and tb is None:
                continue
            segs = ([("v", 0, Mv, vb)] if vb is not None else []) + ([("t", Mv, M, tb)] if tb is not None else [])
            last_live = i == n_last and compact
            # --- ln_1: folded where the previous c_proj produced these rows' operands, a LayerNorm pass elsewhere
            modes = []
            xa_stream = False                               # the image rows' fp32 stream sits in XA until out_proj moves it back to X
            for tower, r0, r1, b in segs:
                src, raw = X[r0:r1], None
                if tower == "v" and i in self.lateral:
                    j = self.lateral.index(i)
                    # the adapter applies ln_1 itself (rows in registers) unless this block's out_proj is not the producing kernel
                    xa_stream = not last_live and not self.fp8 and not self.opt.adapter_ln1_pass
                    ln1 = b["ln1"] if xa_stream else None
                    if conv_events is not None:
                        self._wait(torch.cuda.current_stream(self.dev), conv_events[j])
                        self._adapter(j, w, Bi, t=w["Ts"][j], ln1=ln1)
                    else:
                        self._parallel_stage(j, w, Bi)
                        self._adapter(j, w, Bi, ln1=ln1)
                    if taps is not None:
                        if j:
                            c3 = self.par_specs[j][3]
                            self._tap_nhwc(taps, f"parallel{j}", w["par"][j], Bi, c3.h_out, c3.cout)
                        self._tap_tokens(taps, f"adapter{j}", w["XA"], Bi, self.Lv)
                    src, raw = w["XA"], X[:Mv]              # ln_1 reads the adapter output and moves it back into X
                    pend[tower] = False
                if xa_stream and tower == "v":
                    modes.append((tower, r0, r1, False))
                elif pend[tower]:
                    modes.append((tower, r0, r1, True))
                else:
                    hip.layernorm_stats(src, b["ln1"].g, b["ln1"].b, LNO[r0:r1], r1 - r0, CEN[r0:r1], RST[r0:r1], raw_out=raw,
                                        mdev=self._md(w, r0, r1))
                    modes.append((tower, r0, r1, False))
                pend[tower] = False
            shared = len(segs) == 2 and vb["w"] is tb["w"]
            groups = [(segs[0][1], segs[-1][2], modes)] if shared else [(m[1], m[2], [m]) for m in modes]
            if last_live and not self.opt.last_block_all_queries:
                assert not any(m[3] for m in modes)       # (the block before the last one does not produce: see below)
                cg = [(r0, r1, (self.vblk if ms[0][0] == "v" else self.tblk)[i]["w"]) for r0, r1, ms in groups]
                self._last_block_attention(w, Bi, Bt, cg)
                self._last_block_tail(w, Bi, Bt, vb, tb, attended=True)
                continue
            if shared and not last_live and self.fused_qkv_attn_enabled():
                self._fused_qkv_attention(w, i, groups[0][0], groups[0][1], groups[0][2], Bi, Bt)
            else:
                for r0, r1, ms in groups:
                    self._fold_proj(w, i, "qkv", r0, r1, ms, QKV, hip.ACT_NONE)
                if vb is not None:
                    hip.attention(QKV[:Mv], AO[:Mv], Bi, self.Lv, self.heads, False)
                if tb is not None:
                    self._attention_text(w, QKV, AO, Bt)
            if last_live:
                self._last_block_tail(w, Bi, Bt, vb, tb)
                continue
            # --- out_proj produces ln_2's operands; c_fc consumes them; c_proj produces the next block's ln_1 operands unless
            #     that block takes a LayerNorm pass anyway (the compact last block; image rows in front of a lateral adapter
            #     are produced too -- one launch over both towers -- and overwritten by the adapter's pass)
            nxt_fold = i + 1 <= n_last and not (i + 1 == n_last and compact and not self.opt.last_block_all_queries)
            for r0, r1, ms in groups:
                bw = (self.vblk if ms[0][0] == "v" else self.tblk)[i]["w"]
                md = self._md(w, r0, r1)
                if self.fp8:
                    # out_proj plain (bf16), ln_2 as the e4m3 LayerNorm pass, MLP on the fp8 MFMA; a calibrated c_proj over whole
                    # tiles produces the next block's ln_1 operands like the bf16 one
                    hip.gemm(AO[r0:r1], bw.wo, X[r0:r1], bias=bw.bo, resid=X[r0:r1], resid_kind=hip.RESID_F32, mdev=md)
                    segs2 = [(m[1], m[2], (self.vblk if m[0] == "v" else self.tblk)[i]) for m in ms]
                    self._ln_f8(w, segs2, "ln2")
                    prod = nxt_fold and bw.hid_scale is not None and self._calib is None
                    self._mlp_f8(w, r0, r1, bw, fold_out=hip.FoldOut(LNO[r0:r1], CEN[r0:r1], PART[r0:r1]) if prod else None)
                    if prod:
                        hip.rowstat_finalize(PART[r0:r1], CEN[r0:r1], RST[r0:r1], r1 - r0, D, mdev=md)
                        for m in ms:
                            pend[m[0]] = True
                    continue
                fo = hip.FoldOut(LNO[r0:r1], CEN[r0:r1], PART[r0:r1])
                res = X[r0:r1]
                if xa_stream and ms[0][0] == "v":           # (image rows come first: r0 == 0)
                    res = w["XA"]
                    if r1 > Mv:
                        fo = hip.FoldOut(LNO[r0:r1], CEN[r0:r1], PART[r0:r1], resid2=X[r0:r1], split=Mv)
                hip.gemm(AO[r0:r1], bw.wo, X[r0:r1], bias=bw.bo, resid=res, resid_kind=hip.RESID_F32, fold_out=fo, mdev=md)
                hip.rowstat_finalize(PART[r0:r1], CEN[r0:r1], RST[r0:r1], r1 - r0, D, mdev=md)
                self._fold_proj(w, i, "fc", r0, r1, [(m[0], m[1], m[2], True) for m in ms], HID, hip.ACT_QUICKGELU)
                if nxt_fold:
                    hip.gemm(HID[r0:r1], bw.wpr, X[r0:r1], bias=bw.bpr, resid=X[r0:r1], resid_kind=hip.RESID_F32,
                             fold_out=hip.FoldOut(LNO[r0:r1], CEN[r0:r1], PART[r0:r1]), mdev=md)
                    hip.rowstat_finalize(PART[r0:r1], CEN[r0:r1], RST[r0:r1], r1 - r0, D, mdev=md)
                    for m in ms:
                        pend[m[0]] = True
                else:
                    hip.gemm(HID[r0:r1], bw.wpr, X[r0:r1], bias=bw.bpr, resid=X[r0:r1], resid_kind=hip.RESID_F32, mdev=md)
            if taps is not None:
                if vb is not None:
                    self._tap_tokens(taps, f"vblock{i}", X[:Mv], Bi, self.Lv)
                if tb is not None:
                    self._tap_text(taps, f"tblock{i}", X[Mv:M], w, Bt)

    def _blocks(self, w, Bi, Bt, taps=None, conv_events=None, compact=False, layers=None):
        if self._fold_eligible(w, Bi, Bt):
            return self._blocks_fold(w, Bi, Bt, taps, conv_events, compact, layers)
        Mv, M = w["Mv"], w["M"]
        X, LNO, QKV, AO, HID = w["X"], w["LNO"], w["QKV"], w["AO"], w["HID"]
        for i in (range(self.n_layers) if layers is None else layers):
            vb = self.vblk[i] if Bi else None
            tb = self.tblk[i] if Bt else None
            if vb is None and tb is None:
                continue
            segs = []                      # (row0, row1, block dict)
            if vb is not None:
                segs.append((0, Mv, vb))
            if tb is not None:
                segs.append((Mv, M, tb))
            # --- lateral adapter in front of this vision block
            vis_src = X[:Mv] if vb is not None else None
            raw = None
            if vb is not None and i in self.lateral:
                j = self.lateral.index(i)
                if conv_events is not None:
                    self._wait(torch.cuda.current_stream(self.dev), conv_events[j])
                    self._adapter(j, w, Bi, t=w["Ts"][j])
                else:
                    self._parallel_stage(j, w, Bi)
                    self._adapter(j, w, Bi)
                if taps is not None:
                    if j:
                        c3 = self.par_specs[j][3]
                        self._tap_nhwc(taps, f"parallel{j}", w["par"][j], Bi, c3.h_out, c3.cout)
                    self._tap_tokens(taps, f"adapter{j}", w["XA"], Bi, self.Lv)
                vis_src, raw = w["XA"], X[:Mv]          # ln_1 reads the adapter output and moves it back into X
            # --- ln_1 (modality specific parameters; one launch over both towers' rows unless the adapter output
            #     has to be picked up from its own buffer)
            f8_qkv = self.fp8_qkv
            if f8_qkv:
                if raw is not None:                          # the adapter's output moves back into the residual matrix first
                    hip.gather_rows(vis_src, raw, Mv)
                self._ln_f8(w, segs, "ln1")
            elif len(segs) == 2 and raw is None:
                hip.layernorm_split(X[:M], vb["ln1"].g, vb["ln1"].b, tb["ln1"].g, tb["ln1"].b, Mv, LNO[:M], M)
            else:
                for r0, r1, b in segs:
                    if b is vb:
                        hip.layernorm(vis_src, b["ln1"].g, b["ln1"].b, LNO[r0:r1], r1 - r0, raw_out=raw)
                    else:
                        hip.layernorm(X[r0:r1], b["ln1"].g, b["ln1"].b, LNO[r0:r1], r1 - r0)
            # --- projections: one launch over both towers when the tensors are shared
            groups = [(segs[0][0], segs[-1][1], segs[0][2]["w"])] if len(segs) == 2 and vb["w"] is tb["w"] else \
                     [(r0, r1, b["w"]) for r0, r1, b in segs]
            last_live = i == self.n_layers - 1 and compact
            if last_live and not f8_qkv and not self.opt.last_block_all_queries:
                self._last_block_attention(w, Bi, Bt, groups)
                self._last_block_tail(w, Bi, Bt, vb, tb, attended=True)
                continue
            if (len(groups) == 1 and len(segs) == 2 and not last_live and not f8_qkv and self.fused_qkv_attn_enabled()):
                # opt-in: in_proj + attention of both towers' rows in one kernel (plain weights: LNO holds the LayerNorm outputs)
                self._fused_qkv_attention(w, i, 0, M, [("v", 0, Mv, False)], Bi, Bt)
            else:
                for r0, r1, bw in groups:
                    if f8_qkv:
                        hip.gemm_f8(w["LNQ"][r0:r1], bw.wqkv_q, QKV[r0:r1], w["RS"][r0:r1], bw.wqkv_s, bias=bw.bqkv)
                    else:
                        hip.gemm(LNO[r0:r1], bw.wqkv, QKV[r0:r1], bias=bw.bqkv)
                if vb is not None:
                    hip.attention(QKV[:Mv], AO[:Mv], Bi, self.Lv, self.heads, False)
                if tb is not None:
                    self._attention_text(w, QKV, AO, Bt)
            if last_live:
                self._last_block_tail(w, Bi, Bt, vb, tb)
                continue
            for r0, r1, bw in groups:
                hip.gemm(AO[r0:r1], bw.wo, X[r0:r1], bias=bw.bo, resid=X[r0:r1], resid_kind=hip.RESID_F32)
            if self.fp8:
                self._ln_f8(w, segs, "ln2")
            elif len(segs) == 2:
                hip.layernorm_split(X[:M], vb["ln2"].g, vb["ln2"].b, tb["ln2"].g, tb["ln2"].b, Mv, LNO[:M], M)
            else:
                for r0, r1, b in segs:
                    hip.layernorm(X[r0:r1], b["ln2"].g, b["ln2"].b, LNO[r0:r1], r1 - r0)
            for r0, r1, bw in groups:
                if self.fp8:
                    self._mlp_f8(w, r0, r1, bw)
                else:
                    hip.gemm(LNO[r0:r1], bw.wfc, HID[r0:r1], bias=bw.bfc, act=hip.ACT_QUICKGELU)
                    hip.gemm(HID[r0:r1], bw.wpr, X[r0:r1], bias=bw.bpr, resid=X[r0:r1], resid_kind=hip.RESID_F32)
            if taps is not None:
                if vb is not None:
                    self._tap_tokens(taps, f"vblock{i}", X[:Mv], Bi, self.Lv)
                if tb is not None:
                    self._tap_text(taps, f"tblock{i}", X[Mv:M], w, Bt)

    def _head_image(self, w, Bi, norm=True, compact=False):               # M.py:2685-2690, 2983
        if compact:                                                       # cls rows already sit in XC[:Bi] (_last_block_tail)
            hip.layernorm(w["XC"], self.ln_post.g, self.ln_post.b, w["hv"], Bi)
        else:
            hip.layernorm(w["X"], self.ln_post.g, self.ln_post.b, w["hv"], Bi, row_mul=self.Lv)
        hip.gemm(w["hv"], self.w_vproj, w["fv_raw"])
        if norm:
            hip.l2norm(w["fv_raw"], w["fv"], w["fvb"])

    def _head_text(self, w, Bt, norm=True, compact=False, Bi=0):          # M.py:3057-3077
        if compact:                                                       # EOT rows already sit in XC[Bi:]
            hip.layernorm(w["XC"][Bi:], self.ln_final.g, self.ln_final.b, w["ht"], Bt)
        else:
            hip.layernorm(w["X"], self.ln_final.g, self.ln_final.b, w["ht"], Bt, row_idx=w["eot"])
        hip.gemm(w["ht"], self.w_tproj, w["ft_raw"])
        if norm:
            hip.l2norm(w["ft_raw"], w["ft"], w["ftb"])

    @staticmethod
    def _gather_buf(w, key, local):
        """Persistent [world * B, E] destination of a feature all-gather in the workspace (allocated once per world size)."""
        if not C.comm.collectives:
            return None
        shape = (C.comm.world_size * local.shape[0], local.shape[1])
        buf = w.get(key)
        if buf is None or tuple(buf.shape) != shape:
            buf = w[key] = torch.empty(shape, dtype=local.dtype, device=local.device)
        return buf

    def _native_collectives(self):
        return self.opt.native_collectives and C.native_comm() is not None and (C.comm.collectives or C.native_world() > 1)

    @staticmethod
    def _native_buf(w, key, local, world):
        shape = (world * local.shape[0], local.shape[1])
        buf = w.get(key)
        if buf is None or tuple(buf.shape) != shape:
            buf = w[key] = torch.empty(shape, dtype=local.dtype, device=local.device)
        return buf

    def _heads(self, w, Bi, Bt, norm=True, gather=False, compact=False):
        """Projection heads.  With gather=True the image features' all-gather is started as soon as they exist and
        runs on RCCL's stream while the text head computes (returns the gathered operands and the work handles)."""
        allI = allT = wi = wt = None
        if gather and self._native_collectives():
            # RCCL through the C ABI on the compute stream (comm.init_native_comm): the two gathers are ordinary stream-ordered
            # launches -- entries of the launch plan, capturable -- instead of ProcessGroupNCCL work items on a side stream
            nc, world = C.native_comm(), C.native_world()
            if Bi:
                self._head_image(w, Bi, norm, compact)
                allI = self._native_buf(w, "allI_nat", w["fvb"], world)
                hip.allgather_feats(nc, w["fvb"], allI)
            if Bt:
                self._head_text(w, Bt, norm, compact, Bi)
                allT = self._native_buf(w, "allT_nat", w["ftb"], world)
                hip.allgather_feats(nc, w["ftb"], allT)
            return allI, allT
        if gather and Bi and Bi == Bt and norm and self.opt.gather_packed:
            # SURVEY s8(e)'s form: ONE all-gather of the packed [B, 2, E] unit features after both heads (half the collectives,
            # twice the payload; it cannot start before the text head).  The default below starts the image gather under the
            # text head instead.  Behind a flag so that the first run on a real node can A/B the two.
            E = self.E
            if w.get("pk") is None or w["pk"].shape[0] != Bi:
                w["pk"] = torch.zeros(Bi, 2 * E, dtype=torch.bfloat16, device=self.dev)
                w["fvb_own"], w["ftb_own"] = w["fvb"], w["ftb"]
            w["fvb"], w["ftb"] = w["pk"][:, :E], w["pk"][:, E:]          # the heads' L2 norm writes the halves in place
            self._head_image(w, Bi, norm, compact)
            self._head_text(w, Bt, norm, compact, Bi)
            allpk, h = C.gather_rows_async(w["pk"], out=self._gather_buf(w, "allpk_buf", w["pk"]))
            if h is not None:
                h.wait()
            return allpk[:, :E], allpk[:, E:]
        if w.get("pk") is not None:                                      # (a previous call ran packed: back to the dense buffers)
            w["fvb"], w["ftb"] = w["fvb_own"], w["ftb_own"]
        if Bi:
            self._head_image(w, Bi, norm, compact)
            if gather:
                allI, wi = C.gather_rows_async(w["fvb"], out=self._gather_buf(w, "allI_buf", w["fvb"]))
        if Bt:
            self._head_text(w, Bt, norm, compact, Bi)
            if gather:
                allT, wt = C.gather_rows_async(w["ftb"], out=self._gather_buf(w, "allT_buf", w["ftb"]))
        for h in (wi, wt):
            if h is not None:
                h.wait()                                                  # compute stream waits; the host does not
        return allI, allT

    # ------------------------------------------------------------------ public
    def _check_img(self, img):
        if not img.is_cuda:
            raise hip.HipUnavailable("inputs must be on the HIP device (no CPU path)")
        if img.dim() != 4 or img.shape[1] != 3 or img.shape[2] != self.S or img.shape[3] != self.S:
            raise ValueError(f"expected images [B, 3, {self.S}, {self.S}], got {tuple(img.shape)}")
        if img.dtype not in (torch.float32, torch.bfloat16):
            img = img.float()
        return img.contiguous()

    def _check_tok(self, tok):
        if not tok.is_cuda:
            raise hip.HipUnavailable("inputs must be on the HIP device (no CPU path)")
        if tok.dim() != 2 or tok.shape[1] != self.Lt:
            raise ValueError(f"expected tokens [B, {self.Lt}], got {tuple(tok.shape)}")
        return tok.to(torch.int64).contiguous()

    # per-call state a pass leaves in the workspace (restored before a plan replay: another kind of call may have run in between)
    _CALL_STATE = ("packed", "dyn", "cap", "cu", "len", "Mt_live", "Lmax", "pad", "Mt", "M", "mdev_t", "mdev_all")

    @hip.off_default_stream
    def run(self, img=None, tok=None, norm=True, gather=False, taps=None):
        """Both towers (either may be None) up to the (optionally L2-normalised) features; returns the workspace
        (plus the gathered bf16 features under "allI"/"allT" when gather=True).  `taps` (a dict) receives fp32 copies
        of the intermediate tensors the reference exposes through forward hooks (tests/golden tap_* names).

        The launches of a call depend only on (batch sizes, options, text mode) once the packed row count lives on the device
        (dynamic_rows), so the first call of a kind records them into a native launch table (hip.Plan / msclip_plan_*) while it
        runs, and every later call of that kind replays the table: no per-launch Python, ~2 us of host time per launch."""
        with torch.cuda.device(self.dev):
            capturing = torch.cuda.is_current_stream_capturing()
            if not capturing:
                self.refresh()
            Bi = img.shape[0] if img is not None else 0
            Bt = tok.shape[0] if tok is not None else 0            # (a tensor or a staged Captions batch)
            if self.fp8 and self._calib is None and not self.fp8_calibrated() and not capturing:
                if Bi and Bt:
                    self.calibrate_fp8(img, tok)          # first two-modality batch = the calibration batch (explicit call: calibrate_fp8)
                elif not self._fp8_warned:
                    self._fp8_warned = True
                    import warnings
                    warnings.warn("PRECISION fp8: the MLP hidden scales are not calibrated yet (engine.calibrate_fp8(images, captions)); "
                                  "single-modality calls run c_proj in bf16 until then")
            w = self._workspace(Bi, Bt, inference=True)
            imgc = self._check_img(img) if Bi else None
            cap = tokc = None
            pack = bool(Bt) and self.text_pack_enabled()
            dyn = pack and self.dynamic_rows(Bi, Bt)      # the packed row count stays on the device: nothing below reads the host
            if pack and not dyn and capturing:
                pack = False                                # (the host-read form of packing cannot be captured: full rows)
            if Bt:
                if isinstance(tok, Captions):
                    cap, tokc = (tok if pack else None), tok.tok
                    # a batch staged on a prefetch stream: its token tensor belongs to that stream's pool, this call reads it here
                    tokc.record_stream(torch.cuda.current_stream(self.dev))
                else:
                    tokc = self._check_tok(tok)
            lmax = None        # (device-side row counts: every launch is sized for context_length, whatever the host may know --
            #                     the results do not depend on whether a batch was staged ahead)
            mode = "dyn" if dyn else "pack" if pack else "full"
            key = self._plan_key(Bi, Bt, imgc, mode, lmax, norm, gather, taps)
            self.last_plan = None                           # (the launch table this call ran from, if it did: bench.py's probes)
            if key is not None:
                plan = w.setdefault("plans", {}).get(key)
                streams = [torch.cuda.current_stream(self.dev), self.conv_stream(), _text0_stream(self.dev)]
                ext = [t for t in (imgc, tokc) if t is not None]
                if plan is not None:
                    self.last_plan = plan
                    w.update(plan.state)
                    if tokc is not None:
                        tokc.record_stream(streams[2])          # (the text front may read it there)
                    plan.run(streams, ext)
                    return w
                if not capturing:
                    plan = hip.Plan(self.dev, streams)
                    self._rec = plan
                    try:
                        with plan.recording(externals=ext):
                            self._run_body(w, imgc, tokc, cap, Bi, Bt, mode, lmax, norm, gather, taps)
                    finally:
                        self._rec = None
                    plan.state = {k: w.get(k) for k in self._CALL_STATE}
                    w["plans"][key] = plan
                    self.last_plan = plan
                    return w
            return self._run_body(w, imgc, tokc, cap, Bi, Bt, mode, lmax, norm, gather, taps)

    def _plan_key(self, Bi, Bt, imgc, mode, lmax, norm, gather, taps):
        """What a recorded launch table is valid for, or None when this call cannot be replayed from one: taps (host-side copies),
        an fp8 calibration pass, packed captions sized by a host read, Python-side launch probes, collectives through
        torch.distributed (unless EngineOptions.native_collectives routes them through the C ABI), the opt-in fused kernel's
        per-call tables."""
        if not self.opt.plan or taps is not None or self._calib is not None or (Bt and mode == "pack"):
            return None
        if hip.python_probes_active() or self.fused_qkv_attn_enabled():
            return None
        if gather and C.comm.collectives and not self._native_collectives():
            return None
        return (mode, bool(norm), bool(gather), imgc.dtype if imgc is not None else None, self.opt)

    def drop_plans(self):
        """Forget every recorded launch table (they hold raw addresses of packed weights and workspace buffers)."""
        for w in self._ws.values():
            if isinstance(w, dict):
                w.pop("plans", None)

    def _run_body(self, w, imgc, tokc, cap, Bi, Bt, mode, lmax, norm, gather, taps):
        w["fold_pending"] = {"v": False, "t": False}
        w["fused_tabs"] = None
        conv_events = None
        side_ok = taps is None and self.opt.conv_side_stream and self._multi_stream_ok()
        text0 = ts = None
        if Bt and mode == "full":
            self._text_unpacked(w, Bt)
        if mode == "dyn":
            self._text_lengths_dynamic(tokc, w)
        if Bi and Bt and side_ok and self.vblk[0] is None and self.opt.text0_stream:
            # Text block 0 is text-only (vision slot 0 is the conv stem, M.py:2040-2051) and depends on the captions only:
            # the text front and that block run on a second side stream beside the image front (HBM-bound conv passes
            # beside MFMA-bound projections on disjoint rows / buffers of the workspace); the layer loop waits for it.
            cur = torch.cuda.current_stream(self.dev)
            ts = _text0_stream(self.dev)
            self._wait(ts, self._record(cur))               # the workspace is free: the previous step's work is queued
            tokc.record_stream(ts)
        if mode == "dyn":
            self._text_dynamic(w, Bt, lmax)
        elif mode == "pack":
            # packed captions sized by the host (the round-5 path; small batches and dynamic_rows=False): the total live row count
            # sizes every launch over the text rows.  A batch staged ahead (stage_captions: an input pipeline's prefetch stage)
            # has it ready; otherwise it is staged here and the host waits for its 8 bytes -- i.e. until the previous step's
            # queued work has drained -- BEFORE anything of this call is queued, so that the text front + block 0 still start
            # beside the image front.
            if cap is None:
                with torch.cuda.stream(ts if ts is not None else torch.cuda.current_stream(self.dev)):
                    cap = self.stage_captions(tokc)
            self._text_sizes(cap, w, Bt)
            if ts is not None:                              # (staged on one of the two streams, read on both)
                for t in (cap.len, cap.cu, cap.eot):
                    t.record_stream(ts)
                    t.record_stream(torch.cuda.current_stream(self.dev))

        if ts is not None:
            with torch.cuda.stream(ts):
                self._text_front(tokc, w, Bt)
                self._blocks(w, 0, Bt, layers=(0,))
                text0 = self._record(ts)
        if Bi:
            self._vision_front(imgc, w, Bi, taps)
            # default (EngineOptions.conv_side_stream): +1.6 % pairs/s on B/32, +2.2 % on B/16 same-box.  The GEMM
            # launches it overlaps measure ~11 % longer each, so bench.py takes its per-kernel roofline from a probe
            # pass with the inline schedule and reports the overlapped figure beside it.
            if side_ok and self.lateral and self.lateral == sorted(self.lateral):
                conv_events = self._conv_branch_on_side_stream(w, Bi)
        if Bt and text0 is None:
            self._text_front(tokc, w, Bt)
        # the last block's row-wise tail on the live rows only (EngineOptions.full_last_block: every row, as the taps need it)
        compact = taps is None and not self.opt.full_last_block and not self.lateral_on_last()
        if text0 is not None:
            self._wait(torch.cuda.current_stream(self.dev), text0)
            self._blocks(w, Bi, Bt, taps, conv_events, compact, layers=range(1, self.n_layers))
        else:
            self._blocks(w, Bi, Bt, taps, conv_events, compact)
        allI, allT = self._heads(w, Bi, Bt, norm, gather, compact)
        if gather:
            w["allI"], w["allT"] = allI, allT
        return w

    # ------------------------------------------------------------------ hipGraph replay
    def graph(self, Bi=0, Bt=0, img_dtype=torch.float32):
        """Capture run() for a fixed (Bi, Bt) into a hipGraph (torch.cuda.CUDAGraph records the kernels the C ABI
        launches on the capture stream).  Returns a callable replay(img=None, tok=None) -> workspace: inputs are
        copied into static buffers, the ~230 launches of a step become one graph launch -- what matters for the
        launch-bound small batches of the zero-shot loops (reference tools/zero_shot.py:122-134, 253-275)."""
        key = ("graph", Bi, Bt, img_dtype)
        if key in self._ws:
            return self._ws[key]
        if self.fp8 and not self.fp8_calibrated():
            # the warm-up below runs on an all-zero image and dummy tokens: it must never become the calibration batch
            raise RuntimeError("PRECISION fp8: call engine.calibrate_fp8(images, captions) (or load_fp8_state) before capturing a hipGraph")
        with torch.cuda.device(self.dev):
            simg = torch.zeros(Bi, 3, self.S, self.S, dtype=img_dtype, device=self.dev) if Bi else None
            stok = torch.zeros(Bt, self.Lt, dtype=torch.int64, device=self.dev) if Bt else None
            if Bt:
                stok[:, 0], stok[:, 1] = 1, 2
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                      # warm-up outside capture (workspace allocation, lazy init)
                self.run(simg, stok)
            torch.cuda.current_stream().wait_stream(side)
            self._workspace(Bi, Bt)["pinned"] = True           # the capture bakes this workspace's addresses in
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                w = self.run(simg, stok)

        stamp = self._stamp

        def replay(img=None, tok=None):
            if self._fingerprint() != stamp:
                raise RuntimeError("model tensors changed after this hipGraph was captured (its kernels read the old "
                                   "packed weights): call engine.graph(...) again")
            if Bi:
                simg.copy_(self._check_img(img))
            if Bt:
                stok.copy_(self._check_tok(tok))
            g.replay()
            return w
        self._ws[key] = replay
        return replay

    @hip.off_default_stream
    def encode_image(self, img, norm=True):
        w = self.run(img=img, norm=norm)
        return (w["fv"] if norm else w["fv_raw"]).clone()

    @hip.off_default_stream
    def encode_text(self, tok, norm=True):
        w = self.run(tok=tok, norm=norm)
        return (w["ft"] if norm else w["ft_raw"]).clone()

    def _gathered(self, img, tok, gather):
        if img.shape[0] != tok.shape[0]:
            raise ValueError("forward(image, text) needs the same number of images and captions")
        w = self.run(img, tok, gather=True) if gather else self.run(img, tok)
        if gather:
            return w, w["allI"], w["allT"]
        return w, w["fvb"], w["ftb"]

    @hip.off_default_stream
    def forward_logits(self, img, tok, gather=True):
        """Reference-faithful full N x N logits on every rank (M.py:3136-3141)."""
        w, allI, allT = self._gathered(img, tok, gather)
        n = allI.shape[0]
        out = torch.empty(n, n, dtype=torch.float32, device=self.dev)
        hip.gemm(allI, allT, out, alpha=self.logit_scale_exp)
        return out

    @hip.off_default_stream
    def forward_loss(self, img, tok, gather=True):
        """Symmetric CE over the global batch from the LOCAL row and column blocks only (SURVEY.md s8e option B),
        each as one fused MFMA GEMM + online log-sum-exp sweep (no logits block is written): image rows against all
        captions, caption rows against all images, label logit from the first sweep; the per-rank partial sums are
        all-reduced.  Identical to 0.5*(CE(logits)+CE(logits^T)) of the full matrix."""
        w, allI, allT = self._gathered(img, tok, gather)
        B, n = w["fvb"].shape[0], allI.shape[0]
        world = n // B
        loss = self.loss_from_features(w["fvb"], w["ftb"], allI, allT, C.local_label_offset(B) if world > 1 else 0)
        if world > 1 or (gather and C.comm.collectives):
            if self._native_collectives():
                hip.allreduce(C.native_comm(), loss)
            else:
                dist.all_reduce(loss)
        return loss[0]

    def loss_from_features(self, loc_i, loc_t, all_i, all_t, label_off):
        """This rank's share of the symmetric CE: local unit features [B, E] (bf16) against the rank-major gathered
        ones [N, E]; global label of local row r is label_off + r (reference lib/utils/comm.py:150-153).  Returns a
        1-element tensor; the sum over ranks is the loss.  (forward_loss = towers + gather + this + all-reduce.)"""
        B, n = loc_i.shape[0], all_i.shape[0]
        assert loc_t.shape[0] == B and all_t.shape[0] == n and 0 <= label_off and label_off + B <= n
        ntile = (n + 31) // 32
        nsplit = max(1, min(ntile, 64, 2048 // max(1, (B + 31) // 32)))
        key = ("lse", B, nsplit)
        w = self._ws.setdefault("loss_ws", {})
        if key not in w:
            w[key] = (torch.empty(4, B, nsplit, dtype=torch.float32, device=self.dev),
                      torch.empty(B, dtype=torch.float32, device=self.dev),
                      torch.empty(1, dtype=torch.float32, device=self.dev))
        part, diag, out = w[key]
        s = self.logit_scale_exp
        hip.clip_lse_fused(loc_i, all_t, s, label_off, nsplit, part[0], part[1], diag)
        hip.clip_lse_fused(loc_t, all_i, s, label_off, nsplit, part[2], part[3], diag)
        hip.clip_loss_from_partials(part[0], part[1], part[2], part[3], diag, 1.0 / (2.0 * n), out)
        return out.clone()
