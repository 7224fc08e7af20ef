"""The one-wave-per-(sample, head) attention kernels at the C2 shapes: image (512 x 50 tokens) and packed captions."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msclip_amd import hip, synth  # noqa: E402


def t(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    H, D = 12, 768
    bufs = [torch.randn(512 * 50, 3 * D, device="cuda").to(torch.bfloat16) for _ in range(12)]      # rotated past the MALL
    out = torch.empty(512 * 50, D, dtype=torch.bfloat16, device="cuda")
    i = [0]

    def img():
        i[0] = (i[0] + 1) % len(bufs)
        hip.attention(bufs[i[0]], out, 512, 50, H, False)
    print(f"image attention 512 x 50: {t(img):7.1f} us")
    tok = synth.synth_tokens(512, seed=100).cuda()
    lens = (tok.argmax(-1) + 1).to(torch.int32)
    cu = torch.zeros(514, dtype=torch.int32, device="cuda")
    cu[1:513] = lens.cumsum(0); cu[513] = cu[512]
    total = int(cu[512])
    tb = [torch.randn(total + 256, 3 * D, device="cuda").to(torch.bfloat16) for _ in range(12)]
    to = torch.empty(total + 256, D, dtype=torch.bfloat16, device="cuda")

    def txt():
        i[0] = (i[0] + 1) % len(tb)
        hip.attention_varlen(tb[i[0]], to, cu, 512, 77, H, True)
    print(f"packed caption attention ({total} rows): {t(txt):7.1f} us")


if __name__ == "__main__":
    main()
