"""Times the vocabulary-projection GEMM of every libgemm_*.so next to this file (experiments / ablations)."""
import ctypes, glob, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N, K = 32000, 256
a = torch.randn((M, K), device='cuda'); w = torch.randn((K, N), device='cuda'); b = torch.randn(N, device='cuda')
out = torch.zeros((M, N), device='cuda')
st = torch.cuda.current_stream().cuda_stream
i64, vp = ctypes.c_int64, ctypes.c_void_p
for path in sorted(glob.glob(os.path.join(here, 'libgemm_*.so'))):
    lib = ctypes.CDLL(path)
    f = lib.tst_gemm_mfma_f32
    f.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i64, i64, ctypes.c_int, vp]
    def run(): assert f(a.data_ptr(), K, w.data_ptr(), N, b.data_ptr(), None, 0, out.data_ptr(), N, M, N, K, 0, st) == 0
    for _ in range(2): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print('%-40s M=%d: median %.3f ms %.1f TFLOP/s' % (os.path.basename(path), M, ts[3], 2.0 * M * N * K / ts[3] / 1e9))
