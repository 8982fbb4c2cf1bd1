"""Times the bf16 vocabulary projection of every libbf16_*.so next to this file (experiments / ablations)."""
import ctypes, glob, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
N, K = 32000, 512
st = torch.cuda.current_stream().cuda_stream
i64, vp = ctypes.c_int64, ctypes.c_void_p
for M in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '65536,20480').split(',')]:
    a = torch.randn((M, K), device='cuda').to(torch.bfloat16); w = (torch.randn((N, K), device='cuda') / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device='cuda'); out = torch.zeros((M, N), device='cuda', dtype=torch.bfloat16)
    for path in sorted(glob.glob(os.path.join(here, 'libbf16_*.so'))):
        lib = ctypes.CDLL(path)
        f = lib.tst_gemm_mfma_bf16
        f.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i64, i64, ctypes.c_int, ctypes.c_int, vp]
        def run(): assert f(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), None, 0, out.data_ptr(), N, M, N, K, 0, 1, st) == 0
        for _ in range(2): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort()
        print('%-28s M=%d: median %.3f ms %.0f TFLOP/s  %.2f TB/s out' % (os.path.basename(path), M, ts[3], 2.0 * M * N * K / ts[3] / 1e9, M * N * 2 / ts[3] / 1e9), flush=True)
    del a, w, b, out
