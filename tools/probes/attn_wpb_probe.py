"""Experiment: waves per workgroup (MSCLIP_ATTN_WPB) and workgroups per CU (MSCLIP_ATTN_LDS_KB: dynamic LDS padded to limit occupancy) of
the short-sequence attention kernel at the C2 shapes: image attention (512 x 50 tokens, 12 heads) and packed caption attention (512
captions, U{4..60} + 2 tokens).  Buffers rotate through 5 copies (> the 256 MB MALL) so that the rates are HBM rates."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from msclip_amd import hip, synth

R = 5


def bench(fn, n=200):
    for i in range(20):
        fn(i % R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i % R)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    B, H, D = 512, 12, 768
    torch.manual_seed(0)
    qkv_i = [torch.randn(B * 50, 3 * D, device="cuda").bfloat16() for _ in range(R)]
    out_i = [torch.empty(B * 50, D, device="cuda", dtype=torch.bfloat16) for _ in range(R)]
    tok = synth.synth_tokens(B, seed=100).cuda()
    n = (tok.argmax(-1) + 1).int()
    cu = torch.zeros(B + 2, dtype=torch.int32, device="cuda")
    cu[1:B + 1] = n.cumsum(0)
    cu[B + 1] = n.max()
    tot = int(cu[B])
    qkv_t = [torch.randn(tot + 256, 3 * D, device="cuda").bfloat16() for _ in range(R)]
    out_t = [torch.empty(tot + 256, D, device="cuda", dtype=torch.bfloat16) for _ in range(R)]
    gb_i, gb_t = 4 * B * 50 * D * 2 / 1e9, 4 * tot * D * 2 / 1e9
    for wpb, kb in (("", ""), ("4", "40"), ("4", "54"), ("4", "80"), ("2", ""), ("2", "27"), ("2", "40"), ("2", "54"), ("8", ""), ("8", "80")):
        for k, v in (("MSCLIP_ATTN_WPB", wpb), ("MSCLIP_ATTN_LDS_KB", kb)):
            if v:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
        ti = bench(lambda i: hip.attention(qkv_i[i], out_i[i], B, 50, H, False))
        tt2 = bench(lambda i: hip.attention_varlen(qkv_t[i], out_t[i], cu, B, int(n.max()), H, True))
        tt3 = bench(lambda i: hip.attention_varlen(qkv_t[i], out_t[i], cu, B, 77, H, True))
        print(f"WPB {wpb or 'default(4)':>10} LDS>= {kb or '-':>3} KB: image {ti:6.1f} us ({gb_i / ti * 1e3:5.2f} TB/s)  text NT=2 {tt2:6.1f} us "
              f"({gb_t / tt2 * 1e3:5.2f} TB/s)  text NT=3 {tt3:6.1f} us ({gb_t / tt3 * 1e3:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
