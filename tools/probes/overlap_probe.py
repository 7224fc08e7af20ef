import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "overlap_probe.so"))
vp = ctypes.c_void_p
lib.run_overlap.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, vp, vp, vp]
X = torch.randn(2 ** 27, device="cuda").to(torch.bfloat16)   # 256 MiB, power-of-two element count
clk = torch.zeros(4, dtype=torch.int64, device="cuda")
sink = torch.zeros(4, device="cuda")
iters = 2000
for mode, name in [(1, "MFMA only"), (2, "LDS-DMA only"), (3, "MFMA + LDS-DMA"), (6, "plain loads only"), (7, "MFMA + plain loads"), (10, "buffer LDS-DMA only"), (11, "MFMA + buffer LDS-DMA"), (18, "global LDS-DMA L2-res only"), (19, "MFMA + global LDS-DMA L2-res"), (33, "MFMA+ds_read only"), (43, "MFMA+ds_read + buffer DMA"), (64 + 256, "8 waves MFMA only"), (64, "8 waves MFMA + own global DMA"), (64 + 128, "8 waves MFMA + own buffer DMA")]:
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.run_overlap(X.data_ptr(), X.numel(), iters, mode, sink.data_ptr(), clk.data_ptr(), st)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        lib.run_overlap(X.data_ptr(), X.numel(), iters, mode, sink.data_ptr(), clk.data_ptr(), st)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 5 * 1e3
    nw = 8 if mode & 64 else 4
    tf = 256 * nw * iters * 16 * 2 * 32 * 32 * 16 / us / 1e6 if (mode & 1 or mode & 64) else 0
    tb = 256 * 4 * iters * 4 * 1024 / us / 1e6 if (mode & 2 and not mode & 64) else (256 * 8 * iters * 4 * 1024 / us / 1e6 if (mode & 64 and not mode & 256) else 0)
    c = clk.tolist()
    ghz = [c[i] / max(c[i + 1], 1) * 0.1 for i in (0, 2)]   # wall clock ticks at 100 MHz
    print(f"{name:22s}: {us:9.1f} us   MFMA {tf:7.1f} TF   stream {tb:6.2f} TB/s   shader clock GHz (mfma wave, dma wave): {ghz[0]:.2f} {ghz[1]:.2f}")
    clk.zero_()
