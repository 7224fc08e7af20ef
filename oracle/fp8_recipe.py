"""fp32 torch emulation of this build's fp8 recipe (MODEL.SPEC.PRECISION fp8 / fp8-qkv)  --  TEST INFRASTRUCTURE, NOT PRODUCT.

The reference has no fp8 semantics (BASELINE config C5: "parity unpinned").  What the HIP path is checked against instead is
THIS statement of the recipe (DESIGN.md s9 "The fp8 recipe"), written independently of the kernels: quantise -> dequantise in
torch (torch.float8_e4m3fn is torch's own OCP e4m3 rounding), multiply in fp32.  Everything the recipe does not name is the
fp32 oracle (msclip_oracle.py), unchanged.

Recipe v1
---------
format      OCP e4m3 ("e4m3fn": 4 exponent / 3 mantissa bits, bias 7, no infinities, largest finite value 448), round to
            nearest even, values beyond +-448 saturate to +-448.
weights     one fp32 scale per output channel n:  s_w[n] = max_k |W[n, k]| / 448,  Wq[n, k] = e4m3(W[n, k] / s_w[n]);
            quantised once from the fp32 parameter (in_proj's q rows carry head_dim^-0.5 BEFORE quantisation: M.py:707).
LN-fed      c_fc (and in_proj under fp8-qkv) read the LayerNorm output (M.py:204-219, fp32 statistics) with one fp32 scale
activations per token m:  s_x[m] = max_k |LN(x)[m, k]| / 448,  xq[m, k] = e4m3(LN(x)[m, k] / s_x[m]).
GEMM        y[m, n] = s_x[m] * s_w[n] * sum_k xq[m, k] Wq[n, k] + b[n]    (products of two e4m3 values are exact in fp32; fp32 sum).
MLP hidden  h = QuickGELU(c_fc output) (M.py:222-224) is stored as e4m3 with ONE static scale per layer:
            hq = e4m3(clamp(h / s_h, -448, 448)),  s_h = 1.25 * max |h| over the calibration batch / 448  (both modalities of a
            shared layer, every rank: the maximum);  c_proj:  y = s_h * s_w[n] * sum_k hq[m, k] Wq[n, k] + b[n].
not fp8     attention (q k^T, softmax, p v), out_proj, the conv stem / branch / adapters, patch conv, the last block's live-row
            tail (its few rows run in bf16), heads, logits, loss: bf16 operands in the product, fp32 here.  Why out_proj stays:
            its operand (the attention output) would need its own per-token quantising pass over the token matrix, which costs
            more than the fp8 main loop of a K = d GEMM saves.
"""
import torch
import torch.nn.functional as F

from oracle import msclip_oracle as O

E4M3_MAX = 448.0
F8 = torch.float8_e4m3fn


def e4m3(x):
    """Round to OCP e4m3 (saturating) and back to fp32."""
    return x.float().clamp(-E4M3_MAX, E4M3_MAX).to(F8).float()


def quant_rows(x):
    """-> (dequantisable e4m3 values as fp32, per-row scale): x ~= q * s[:, None]."""
    s = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / E4M3_MAX
    return e4m3(x / s), s


def linear_f8(x, w, b):
    """Per-token x per-channel scaled e4m3 GEMM of the recipe: x [..., K] (a LayerNorm output), w [N, K]."""
    xq, sx = quant_rows(x)
    wq, sw = quant_rows(w)
    return (xq @ wq.t()) * sx * sw.t() + b


def make_block_fn(precision, hid_scales=None, record=None, live_tail=None):
    """residual_block of the oracle under `precision` ("fp8": c_fc / c_proj; "fp8-qkv": in_proj as well).
    hid_scales: {block prefix: s_h}; a block without one keeps an fp32 hidden matrix (the product's uncalibrated state).
    record: dict that receives {block prefix: max |hidden|} (what calibration measures).
    live_tail: set of block prefixes whose MLP runs unquantised (the last block: the product runs its live rows in bf16)."""
    assert precision in ("fp8", "fp8-qkv")

    def block(x, sd, p, heads, mask=None):
        h1 = O.layer_norm(x, sd[p + ".ln_1.weight"], sd[p + ".ln_1.bias"])
        if precision == "fp8-qkv":
            B, L, C = h1.shape
            hd = C // heads
            w = sd[p + ".attn.in_proj_weight"].clone()
            bq = sd[p + ".attn.in_proj_bias"].clone()
            w[:C] *= float(hd) ** -0.5                       # q pre-scaled before quantisation (the packed weight)
            bq[:C] *= float(hd) ** -0.5
            qkv = linear_f8(h1, w, bq)
            q, k, v = qkv.chunk(3, dim=-1)
            q, k, v = (t.reshape(B, L, heads, hd).transpose(1, 2) for t in (q, k, v))
            s = q @ k.transpose(-1, -2)
            if mask is not None:
                s = s + mask
            o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, C)
            a = F.linear(o, sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
        else:
            a = O.attention(h1, sd, p + ".attn", heads, mask)
        x = x + a
        h2 = O.layer_norm(x, sd[p + ".ln_2.weight"], sd[p + ".ln_2.bias"])
        if live_tail and p in live_tail:
            return x + O.mlp(h2, sd, p + ".mlp")
        hid = O.quick_gelu(linear_f8(h2, sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"]))
        if record is not None:
            record[p] = max(record.get(p, 0.0), float(hid.abs().max()))
        s_h = (hid_scales or {}).get(p)
        if s_h is None:
            return x + F.linear(hid, sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])
        hq = e4m3(hid / s_h)
        wq, sw = quant_rows(sd[p + ".mlp.c_proj.weight"])
        return x + (hq @ wq.t()) * (s_h * sw.t()) + sd[p + ".mlp.c_proj.bias"]
    return block
