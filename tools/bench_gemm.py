"""Microbenchmark of tst_gemm_mfma_f32 on the decode shapes (HIP events, random data, several rounds).
Usage: python tools/bench_gemm.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tst_amd
from tst_amd import kernels as K

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
shapes = [('vocab', rows, 32000, 256, False, 0), ('qkv', rows, 768, 256, False, 0), ('proj+res', rows, 256, 256, True, 0),
          ('ff1 relu', rows, 1024, 256, False, 1), ('ff2+res', rows, 256, 1024, True, 0), ('gru gates', rows, 3072, 1024, False, 0)]
if len(sys.argv) > 2:
    shapes = [s for s in shapes if s[0] in sys.argv[2].split(',')]
g = torch.Generator(device='cuda').manual_seed(0)
for name, M, N, Kd, res, act in shapes:
    a = torch.randn((M, Kd), device='cuda', generator=g)
    w = torch.randn((Kd, N), device='cuda', generator=g)
    b = torch.randn(N, device='cuda', generator=g)
    out = torch.empty((M, N), device='cuda')
    r = torch.randn((M, N), device='cuda', generator=g) if res else None
    for _ in range(2):
        K.gemm(a, w, b, out, M, res=r, act=act)
    torch.cuda.synchronize()
    times = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.gemm(a, w, b, out, M, res=r, act=act)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    times.sort()
    fl = 2.0 * M * N * Kd
    print('%-10s M=%d N=%d K=%d  median %.3f ms  %.1f TFLOP/s (min %.3f ms %.1f)  checksum %.6e'
          % (name, M, N, Kd, times[3], fl / times[3] / 1e9, times[0], fl / times[0] / 1e9, float(out.double().sum())))
    del a, w, b, out, r
