"""Microbenchmark of tst_gemm_mfma_bf16 on the shapes of the bf16 language model (HIP events, random data).
Usage: python tools/bench_gemm_bf16.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tst_amd
from tst_amd import kernels as K

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
shapes = [('vocab', rows, 32000, 512, False, 0, True), ('qkv', rows, 1536, 512, False, 0, True),
          ('proj+res', rows, 512, 512, True, 0, False), ('ff1 relu', rows, 2048, 512, False, 1, True),
          ('ff2+res', rows, 512, 2048, True, 0, False), ('sq8k', 8192, 8192, 8192, False, 0, True),
          ('k4096', 8192, 8192, 4096, False, 0, True), ('k1024', 16384, 8192, 1024, False, 0, True)]
if len(sys.argv) > 2:
    shapes = [s for s in shapes if s[0] in sys.argv[2].split(',')]
g = torch.Generator(device='cuda').manual_seed(0)
for name, M, N, Kd, res, act, obf in shapes:
    a = torch.randn((M, Kd), device='cuda', generator=g).to(torch.bfloat16)
    w = (torch.randn((N, Kd), device='cuda', generator=g) / Kd ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device='cuda', generator=g)
    out = torch.empty((M, N), device='cuda', dtype=torch.bfloat16 if obf else torch.float32)
    r = torch.randn((M, N), device='cuda', generator=g) if res else None
    for _ in range(2):
        K.gemm_bf16(a, w, b, out, M, res=r, act=act)
    torch.cuda.synchronize()
    times = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.gemm_bf16(a, w, b, out, M, res=r, act=act)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    times.sort()
    fl = 2.0 * M * N * Kd
    byts = (M * Kd + N * Kd) * 2 + M * N * (2 if obf else 4) * (2 if res else 1)
    med = times[len(times) // 2]
    print('%-9s M=%d N=%d K=%d  median %.3f ms  %.0f TFLOP/s  %.2f TB/s (min %.3f ms %.0f TF)  checksum %.6e'
          % (name, M, N, Kd, med, fl / med / 1e9, byts / med / 1e9, times[0], fl / times[0] / 1e9, float(out.double().sum())))
    del a, w, b, out, r
