"""How much does the C2 step time depend on the (data-dependent) packed row count?  The persistent 256 x 256 GEMMs run ceil(tiles / 256)
rounds: out_proj / c_proj have 3 column tiles, so 170 row tiles are 1.99 rounds and 171 are 2.004 -> 3.  Times the forward step for
caption batches of different totals (same batch size, lengths shifted)."""
import json
import os
import sys
import time

os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from bench import load_schema
    from msclip_amd import synth
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config
    name, B = "b32-yfcc-msclips", 512
    m = get_clip_model(named_config(name))
    m.load_state_dict(synth.synth_state_dict(load_schema(name), seed=0), strict=True)
    m = m.cuda().eval()
    eng = m.engine()
    img = synth.synth_images(B, seed=10).cuda()
    out = []
    for lo, hi in ((4, 60), (4, 56), (4, 58), (5, 60), (6, 60), (4, 62), (6, 62), (8, 62), (4, 66), (10, 66)):
        tok = synth.synth_tokens(B, seed=100, min_len=lo, max_len=hi).cuda()
        total = int((tok.argmax(-1) + 1).sum())
        rows = 25600 + -(-total // 256) * 256
        for _ in range(4):
            eng.forward_loss(img, tok, gather=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            eng.forward_loss(img, tok, gather=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        out.append({"lengths": [lo, hi], "text_rows": total, "row_tiles": rows // 256, "rounds_N768": round(rows // 256 * 3 / 256, 3),
                    "rounds_N2304": round(rows // 256 * 9 / 256, 3), "rounds_N3072": round(rows // 256 * 12 / 256, 3), "ms": round(ms, 3),
                    "us_per_row_tile": round(ms * 1e3 / (rows // 256), 2)})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
