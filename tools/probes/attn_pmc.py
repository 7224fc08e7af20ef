"""Workload for `rocprofv3 --pmc ...`: the short-sequence attention kernels at the C2 shapes, rotating buffers (HBM-resident data).
    cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
        SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d out -o run -- python tools/probes/attn_pmc.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from msclip_amd import hip, synth

B, H, D, R = 512, 12, 768, 5
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
for i in range(10):
    hip.attention(qkv_i[i % R], out_i[i % R], B, 50, H, False)
    hip.attention_varlen(qkv_t[i % R], out_t[i % R], cu, B, 77, H, True)
torch.cuda.synchronize()
