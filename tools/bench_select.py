"""Microbenchmark of the HBM-bound selection kernels on [rows, 32000] fp32 logits (HIP events, random data).
Usage: python tools/bench_select.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tst_amd
from tst_amd import kernels as K

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
V = 32000
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn((rows, V), device='cuda', generator=g) * 2
tok = torch.empty(rows, dtype=torch.int32, device='cuda')
lse = torch.empty(rows, device='cuda')
sent = torch.arange(rows, dtype=torch.int32, device='cuda')


def timeit(fn, n=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


gb = rows * V * 4 / 1e9
ms = timeit(lambda: K.argmax_rows(x, rows, V, tok))
xb = x.to(torch.bfloat16)
print('argmax_rows          %.3f ms  %.2f TB/s' % (ms, gb / ms))
for k in (1, 4, 8):
    tv = torch.empty((rows, k), device='cuda')
    ti = torch.empty((rows, k), dtype=torch.int32, device='cuda')
    ms = timeit(lambda: K.logsoftmax_topk(x, rows, V, k, tv, ti, lse))
    print('logsoftmax_topk k=%d  %.3f ms  %.2f TB/s' % (k, ms, gb / ms))
ms = timeit(lambda: K.sample_rows(x, rows, V, 1.0, 1, 0, sent, tok))
print('sample_rows          %.3f ms  %.2f TB/s' % (ms, gb / ms))
y = torch.empty_like(x)
ms = timeit(lambda: y.copy_(x))
print('torch copy (r+w)     %.3f ms  %.2f TB/s (read+write bytes)' % (ms, 2 * gb / ms))
for k in (4, 8):
    tv = torch.empty((rows, k), device='cuda')
    ti = torch.empty((rows, k), dtype=torch.int32, device='cuda')
    ms = timeit(lambda: K.logsoftmax_topk(xb, rows, V, k, tv, ti, lse))
    print('logsoftmax_topk bf16 k=%d  %.3f ms  %.2f TB/s' % (k, ms, gb / 2 / ms))
