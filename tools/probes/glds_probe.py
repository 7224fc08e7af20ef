import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "glds_probe.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.run_probe.argtypes = [ci, ci, vp, vp, ci, ci, ci, vp, ci, vp]
M = 65024
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for (N, K) in [(2304, 768), (768, 3072)]:
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    tiles = (M // 256) * (N // 256)
    nbytes = tiles * (K // 64) * 65536
    for depth in (1, 2):
        for pat in (0, 1, 2):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(2):
                lib.run_probe(pat, depth, X.data_ptr(), W.data_ptr(), M, N, K, sink.data_ptr(), 256, st)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                lib.run_probe(pat, depth, X.data_ptr(), W.data_ptr(), M, N, K, sink.data_ptr(), 256, st)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 10 * 1e3
            print(f"N={N} K={K} depth={depth} pattern={pat}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s into LDS")
