"""The vocabulary projection [rows, 256] x [256, 32000] on the weights-in-registers kernel (tst_gemm_mfma_f32 under TST_GEMM_WREG=1)
against the LDS-DMA kernel of the decode step (tst_gemm_dma_f32): time per launch, TFLOP/s, equal bits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tst_amd
from tst_amd import kernels as K

V, D = 32000, 256
g = torch.Generator(device='cuda').manual_seed(0)
w = torch.randn((D, V), device='cuda', generator=g) / 16
b = torch.randn(V, device='cuda', generator=g)


def timeit(fn, n=10):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


timeit(lambda: K.gemm(torch.zeros((4096, D), device='cuda'), w, b, torch.empty((4096, V), device='cuda'), 4096), 20)   # clocks up
for M in [int(a) for a in sys.argv[1:]] or [15676, 32768, 8192, 4096]:
    a = torch.randn((M, D), device='cuda', generator=g)
    at = torch.zeros((D, (M + 127) // 128 * 128), device='cuda')
    at[:, :M] = a.t()
    o1 = torch.empty((M, V), device='cuda')
    o2 = torch.empty((M, V), device='cuda')
    t1 = timeit(lambda: K.gemm(a, w, b, o1, M))
    t2 = timeit(lambda: K.gemm_dma(at, w, b, o2, M, n_cols=V, variant=3))
    flop = 2.0 * M * V * D
    print('rows %6d: tst_gemm_mfma_f32 %8.1f us %6.1f TF | tst_gemm_dma_f32 %8.1f us %6.1f TF | equal %s' % (
        M, t1 * 1e3, flop / t1 / 1e9, t2 * 1e3, flop / t2 / 1e9, bool(torch.equal(o1, o2))), flush=True)
