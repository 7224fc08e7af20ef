import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "mfma_probe.so"))
vp = ctypes.c_void_p
lib.run_mfma.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp]
sink = torch.zeros(4, device="cuda")
iters = 4000
for data in ("random", "zeros"):
    X = (torch.randn(2 ** 26, device="cuda") if data == "random" else torch.zeros(2 ** 26, device="cuda")).to(torch.bfloat16)
    for shape, name in [(0, "32x32x16"), (1, "16x16x32"), (0, "32x32x16"), (1, "16x16x32")]:
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            lib.run_mfma(X.data_ptr(), iters, shape, sink.data_ptr(), st)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            lib.run_mfma(X.data_ptr(), iters, shape, sink.data_ptr(), st)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 10 * 1e3
        tf = 256 * 8 * iters * 16 * 2 * 32 * 32 * 16 / us / 1e6
        print(f"{data:6s} {name}: {us:9.1f} us  {tf:7.1f} TF")
