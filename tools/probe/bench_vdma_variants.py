"""Times tst_gemm_dma_f32 (variant 3) of every libvdma_*.so next to this file on the vocabulary projection."""
import ctypes, glob, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
st = torch.cuda.current_stream().cuda_stream
i64, vp = ctypes.c_int64, ctypes.c_void_p
N, K = 32000, 256
w = torch.randn((K, N), device='cuda') / 16; b = torch.randn(N, device='cuda')
for M in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '32768,15676,8192').split(',')]:
    ld = (M + 127) // 128 * 128
    at = torch.randn((K, ld), device='cuda'); out = torch.zeros((M, N), device='cuda')
    for path in sorted(glob.glob(os.path.join(here, 'libvdma_*.so'))):
        lib = ctypes.CDLL(path)
        f = lib.tst_gemm_dma_f32
        f.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, ctypes.c_int, vp]
        def run(): assert f(at.data_ptr(), ld, w.data_ptr(), N, b.data_ptr(), out.data_ptr(), N, M, N, K, 3, st) == 0
        for _ in range(3): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(11):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort()
        print('M=%d %-24s %.3f ms %.1f TF  checksum %.6e' % (M, os.path.basename(path), ts[5], 2.0 * M * N * K / ts[5] / 1e9, float(out[:64].double().sum())), flush=True)
    del at, out
