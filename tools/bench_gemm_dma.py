"""Vocabulary projection [rows, 256] x [256, 32000]: the LDS-DMA kernel (k-major activations) against gemm_f32_kernel, results
compared bit for bit.  Usage: python tools/bench_gemm_dma.py [rows ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tst_amd import kernels as K


def med(fn, reps=7):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


g = torch.Generator(device='cuda').manual_seed(0)
N, Kd = 32000, 256
w = torch.randn((Kd, N), device='cuda', generator=g) / 16
b = torch.randn(N, device='cuda', generator=g)
for M in [int(a) for a in sys.argv[1:]] or [32768, 15676, 8192, 4096, 2048]:
    a = torch.randn((M, Kd), device='cuda', generator=g)
    ld = (M + 255) // 256 * 256
    at = torch.zeros((Kd, ld), device='cuda')
    at[:, :M] = a.t()
    ref = torch.empty((M, N), device='cuda')
    t0 = med(lambda: K.gemm(a, w, b, ref, M))
    fl = 2.0 * M * N * Kd
    line = 'M=%d  register-staged %.3f ms %.1f TF' % (M, t0, fl / t0 / 1e9)
    for v in (2, 3, 4, 5):
        out = torch.zeros((M, N), device='cuda')
        t1 = med(lambda: K.gemm_dma(at, w, b, out, M, variant=v))
        line += ' | dma v%d %.3f ms %.1f TF %s' % (v, t1, fl / t1 / 1e9, 'same bits' if torch.equal(out, ref) else 'DIFFERENT')
        del out
    print(line, flush=True)
    del a, at, ref
