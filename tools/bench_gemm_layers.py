"""Microbenchmark of tst_gemm_mfma_f32 on the layer products of the transformer-mt base model (HIP events, random data).
Usage: [REPS=n] python tools/bench_gemm_layers.py [rows ...]   (REPS launches between the two events of a timing: 1 includes the
cost of starting a launch on an idle GPU, 10 or more show the kernel as it runs inside a step)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tst_amd
from tst_amd import kernels as K

SHAPES = [('o-proj +res', 256, 256, True, 0), ('cross-q', 256, 256, False, 0), ('qkv', 768, 256, False, 0),
          ('ff1 relu', 1024, 256, False, 1), ('ff2 +res', 256, 1024, True, 0)]


REPS = int(os.environ.get('REPS', '1'))


def med(fn, n=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / REPS)
    return sorted(ts)[len(ts) // 2]


g = torch.Generator(device='cuda').manual_seed(0)
for M in [int(a) for a in sys.argv[1:]] or [15676, 8192, 30000]:
    tot = 0.0
    for name, N, Kd, res, act in SHAPES:
        a = torch.randn((M, Kd), device='cuda', generator=g)
        w = torch.randn((Kd, N), device='cuda', generator=g) / 16
        b = torch.randn(N, device='cuda', generator=g)
        r = torch.randn((M, N), device='cuda', generator=g) if res else None
        out = torch.empty((M, N), device='cuda')
        t = med(lambda: K.gemm(a, w, b, out, M, res=r, act=act))
        tot += t
        print('M=%-6d %-12s N=%-5d K=%-5d %7.1f us %6.1f TF  checksum %.6e' % (M, name, N, Kd, t * 1e3, 2.0 * M * N * Kd / t / 1e9, float(out.double().sum())), flush=True)
    print('M=%-6d sum of the five %.1f us' % (M, tot * 1e3), flush=True)
