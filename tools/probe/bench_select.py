"""Probe: bandwidth of the selection kernel (log-sum-exp + top-k over V = 32000) on fp32 and bf16 rows, median of 15 launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tst_amd
from tst_amd import kernels as K
V = 32000
for rows, k, dt in ((16384, 4, torch.float32), (8192, 4, torch.float32), (20480, 8, torch.bfloat16)):
    x = (torch.randn((rows, V), device='cuda') * 2).to(dt)
    tv = torch.empty((rows, k), device='cuda'); ti = torch.empty((rows, k), dtype=torch.int32, device='cuda'); lse = torch.empty(rows, device='cuda')
    ts = []
    for i in range(18):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); K.logsoftmax_topk(x, rows, V, k, tv, ti, lse); e1.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(e0.elapsed_time(e1))
    ts.sort()
    print('%s rows %d k %d: %.1f us  %.2f TB/s  checksum %.6e %d' % (str(dt)[6:], rows, k, ts[len(ts) // 2] * 1e3, rows * V * x.element_size() / ts[len(ts) // 2] / 1e9,
                                                            float(lse.double().sum()), int(ti.long().sum())), flush=True)
    del x
