"""Per-CU / chip-wide rate of the producer epilogue's memory pattern (tools/probes/epi_probe.hip): GB/s per CU and TB/s total
against the number of workgroups (= CUs) that run it at once.    python tools/probes/epi_probe.py"""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "epi_probe.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.run_epi.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
N = 768
ntn = N // 256
for ntm in (172,):
    M = ntm * 256
    x = torch.randn(M, N, device="cuda")
    xb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for (mp, depth, nt, bf) in [(0, 2, 1, 1), (0, 4, 1, 1), (0, 8, 1, 1), (1, 2, 1, 1), (1, 8, 1, 1), (0, 2, 0, 1), (0, 8, 0, 1), (1, 8, 0, 1), (0, 2, 1, 0), (0, 8, 1, 0), (1, 8, 1, 0)]:
        line = f"map {mp} depth {depth} nt {nt} bf16copy {bf}:"
        for grid in (8, 32, 64, 128, 256):
            # every workgroup processes the same number of tiles (2): tiles = 2 * grid (a sub-matrix of the first rows)
            tm = 2 * grid // ntn
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                assert lib.run_epi(x.data_ptr(), xb.data_ptr(), N, tm, ntn, grid, mp, depth, nt, bf, st) == 0
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                lib.run_epi(x.data_ptr(), xb.data_ptr(), N, tm, ntn, grid, mp, depth, nt, bf, st)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 20 * 1e3
            nbytes = tm * ntn * 256 * 256 * (8 + (2 if bf else 0))
            line += f"  {grid:3d} WG: {us:6.1f} us {nbytes / us / 1e3 / grid:6.1f} GB/s/CU {nbytes / us / 1e6:5.2f} TB/s |"
        print(line)
