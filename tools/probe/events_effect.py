"""Probe: cost of timing launches with HIP events that are created inside the measured batches, against the pre-made
pool of ``tst_amd.timing.LaunchTimer``.  Runs whole batches of the benchmark workload in ONE process and prints, per
batch, the wall time and (read only after the last batch) the selection pass's GB/s.

POOL=0: a new pair of events per launch, all kept alive to the end (how the benchmark used to time).
POOL=1: LaunchTimer.  GC=0 switches the Python collector off during the batches."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from tst_amd.timing import LaunchTimer


class FreshEvents:
    def __init__(self):
        self.spans = []

    def begin(self):
        self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def end(self, work):
        self.e1.record()
        self.spans.append((self.e0, self.e1, work))

    def totals(self):
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b, _ in self.spans), sum(w for _, _, w in self.spans), len(self.spans)


def main():
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    from tst_amd import synthetic
    model, vocab = B.build_model(dev, 5.5)
    src = synthetic.multi30k_like_sources(8192, B.VOCAB, seed=1236).to(dev)
    n = int(os.environ.get('BATCHES', '8'))
    pool = os.environ.get('POOL', '1') == '1'
    mk = (lambda: LaunchTimer(64)) if pool else FreshEvents
    timers = [(mk(), mk()) for _ in range(n)]
    B.decode_batch(model, vocab, src, 4, 0.0, None)
    if os.environ.get('GC', '1') == '0':
        gc.collect()
        gc.disable()
    wall = []
    for g, s in timers:
        model.decoder.event_log = g
        torch.cuda.synchronize()
        t0 = time.time()
        tok, lens, loop = B.decode_batch(model, vocab, src, 4, 0.0, s)
        tok.cpu(), lens.cpu()
        torch.cuda.synchronize()
        wall.append((time.time() - t0) * 1e3)
    for w, (g, s) in zip(wall, timers):
        ms, by, k = s.totals()
        gms, fl, _ = g.totals()
        print('batch %.1f ms  select %.0f GB/s (%.1f ms, %d launches)  projection %.1f TFLOP/s' % (w, by / ms / 1e6, ms, k, fl / gms / 1e9))


main()
