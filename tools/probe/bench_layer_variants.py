"""Times the layer products of every liblayer_*.so next to this file (experiments / ablations)."""
import ctypes, glob, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
st = torch.cuda.current_stream().cuda_stream
i64, vp = ctypes.c_int64, ctypes.c_void_p
M = int(sys.argv[1]) if len(sys.argv) > 1 else 15676
for name, N, K, res in (('o-proj +res', 256, 256, True), ('cross-q', 256, 256, False), ('qkv', 768, 256, False), ('ff2 +res', 256, 1024, True)):
    a = torch.randn((M, K), device='cuda'); w = torch.randn((K, N), device='cuda') / 16; b = torch.randn(N, device='cuda')
    r = torch.randn((M, N), device='cuda') if res else None
    out = torch.zeros((M, N), device='cuda')
    for path in sorted(glob.glob(os.path.join(here, 'liblayer_*.so'))):
        lib = ctypes.CDLL(path)
        f = lib.tst_gemm_mfma_f32
        f.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i64, i64, ctypes.c_int, vp]
        def run(): assert f(a.data_ptr(), K, w.data_ptr(), N, b.data_ptr(), r.data_ptr() if res else None, N if res else 0, out.data_ptr(), N, M, N, K, 0, st) == 0
        for _ in range(3): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort()
        print('M=%d %-12s %-28s %.1f us %.1f TF' % (M, name, os.path.basename(path), ts[7] * 1e3, 2.0 * M * N * K / ts[7] / 1e9), flush=True)
