"""Experiment: vocabulary projection + selection of R rows as one pair of launches or in row chunks small enough
for the logits of a chunk to stay in the memory-side cache between the two kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tst_amd
from tst_amd import kernels as K

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
V, d, k = 32000, 256, 4
g = torch.Generator(device='cuda').manual_seed(0)
h = torch.randn((R, d), device='cuda', generator=g)
w = torch.randn((d, V), device='cuda', generator=g) * 0.1
b = torch.randn(V, device='cuda', generator=g)
logits = torch.empty((R, V), device='cuda')
tv = torch.empty((R, k), device='cuda'); ti = torch.empty((R, k), dtype=torch.int32, device='cuda'); lse = torch.empty(R, device='cuda')

def run(chunk):
    for lo in range(0, R, chunk):
        n = min(chunk, R - lo)
        K.gemm(h[lo:], w, b, logits[lo:], n, n_cols=V)
        K.logsoftmax_topk(logits[lo:], n, V, k, tv[lo:], ti[lo:], lse[lo:])

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]

ref = None
for chunk in (R, 8192, 4096, 2048, 1024, 512):
    if chunk > R: continue
    ms = timeit(lambda: run(chunk))
    chk = (float(lse.double().sum()), int(ti.long().sum()))
    if ref is None: ref = chk
    print('chunk %6d rows: %.3f ms  same=%s' % (chunk, ms, chk == ref))
