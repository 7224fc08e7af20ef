"""Run a few launches of one GEMM shape for counter collection (GPU box only): python tools/gemm_pmc.py <tile> <shape>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import SHAPES, run  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 2
which = sys.argv[2] if len(sys.argv) > 2 else "proj"
for name, M, N, K, epi in SHAPES:
    if name.strip() == which:
        us, tf = run(name, M, N, K, epi, tile, iters=3)
        print(name, us, tf)
