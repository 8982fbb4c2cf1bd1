"""Probe: bandwidth of the selection kernel over the first minute on a fresh box (one measurement per ~2 s)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tst_amd
from tst_amd import kernels as K
rows, V, k = 16384, 32000, 4
x = torch.randn((rows, V), device='cuda') * 2
tv = torch.empty((rows, k), device='cuda'); ti = torch.empty((rows, k), dtype=torch.int32, device='cuda'); lse = torch.empty(rows, device='cuda')
t0 = time.time()
busy = len(sys.argv) > 1 and sys.argv[1] == 'busy'
a = torch.randn((8192, 8192), device='cuda'); 
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 45:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.logsoftmax_topk(x, rows, V, k, tv, ti, lse); e1.record(); torch.cuda.synchronize()
    print('t=%5.1f s  %.2f TB/s' % (time.time() - t0, rows * V * 4 / e0.elapsed_time(e1) / 1e9), flush=True)
    if busy:
        t1 = time.time()
        while time.time() - t1 < 2.0:
            K.logsoftmax_topk(x, rows, V, k, tv, ti, lse)
        torch.cuda.synchronize()
    else:
        time.sleep(2.0)
