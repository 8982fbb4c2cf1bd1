"""Small-M fp32 GEMMs: the 64 x 64-tile kernel against the one-wave-per-strip kernel (strip = 1, 2, 4), checked
bit for bit against each other and timed (HIP events around 20 back-to-back launches, best of 5).
Usage: python tools/bench_gemm_small.py [rows,rows,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tst_amd  # noqa: F401
from tst_amd import kernels as K

rows_list = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [64, 128, 256, 512, 1024, 2048, 4096, 8192]
shapes = [('qkv', 768, 256, False, 0), ('proj+res', 256, 256, True, 0), ('ff1 relu', 1024, 256, False, 1), ('ff2+res', 256, 1024, True, 0),
          ('gru gates', 3072, 1024, False, 0), ('lm qkv', 1536, 512, False, 0), ('lm ff2', 512, 2048, True, 0), ('vocab', 32000, 256, False, 0)]
g = torch.Generator(device='cuda').manual_seed(0)


def timed(fn):
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    return best * 1e3


for name, N, Kd, res, act in shapes:
    for M in rows_list:
        if name == 'vocab' and M > 2048:
            continue
        a = torch.randn((M, Kd), device='cuda', generator=g)
        w = torch.randn((Kd, N), device='cuda', generator=g)
        b = torch.randn(N, device='cuda', generator=g)
        r = torch.randn((M, N), device='cuda', generator=g) if res else None
        ref = torch.empty((M, N), device='cuda')
        K.gemm(a, w, b, ref, M, res=r, act=act)
        line = '%-9s M=%-5d N=%-5d K=%-4d  auto %6.1f us' % (name, M, N, Kd, timed(lambda: K.gemm(a, w, b, ref, M, res=r, act=act)))
        for strip in (1, 2, 4):
            out = torch.full((M, N), float('nan'), device='cuda')
            K.gemm(a, w, b, out, M, res=r, act=act, strip=strip)
            same = bool(torch.equal(out, ref))
            line += ' | strip %d %6.1f us %s' % (strip, timed(lambda: K.gemm(a, w, b, out, M, res=r, act=act, strip=strip)), 'same' if same else 'DIFFERENT')
        print(line, flush=True)
        del a, w, b, r, ref, out
