"""Probe: cost of a config-2 decode step as a function of its row count, time = a + b * rows (least squares over the steps of
a few batches), and from it what keeping the on-GPU batch full would buy on a corpus walk (config 5): the corpus then takes
row_steps / (batch * beam) steps instead of max_len steps per batch, and each step saved saves the fixed part a."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, tst_amd  # noqa: E402
from tst_amd import synthetic, transformer  # noqa: E402

dev = torch.device('cuda', 0)
model, vocab = bench.build_model(dev, 5.5)
log = []
orig = transformer.TransformerSession.step


def step(self, rows, R, t, **kw):
    torch.cuda.synchronize()
    log.append((time.perf_counter(), R, t))
    return orig(self, rows, R, t, **kw)


transformer.TransformerSession.step = step
pairs = []
for rep in range(4):
    src = synthetic.multi30k_like_sources(8192, bench.VOCAB, seed=1236 + rep).to(dev)
    log.clear()
    tok, lens, loop = bench.decode_batch(model, vocab, src, 4, 0.0)
    torch.cuda.synchronize()
    if rep == 0:
        continue
    for (t0, R, t), (t1, _, _) in zip(log[:-1], log[1:]):
        if t >= 1:
            pairs.append((R, (t1 - t0) * 1e6))
    steps, row_steps = loop.steps_done, loop.rows_stepped
R = np.array([p[0] for p in pairs], dtype=np.float64)
us = np.array([p[1] for p in pairs])
A = np.stack([np.ones_like(R), R], 1)
(a, b), res, _, _ = np.linalg.lstsq(A, us, rcond=None)
print('steps fitted: %d; time per step = %.1f us + %.4f us per row (rms residual %.1f us)' % (len(pairs), a, b, np.sqrt(np.mean((A @ [a, b] - us) ** 2))))
for lo, hi in ((0, 4096), (4096, 12288), (12288, 24576), (24576, 40000)):
    m = (R >= lo) & (R < hi)
    if m.any():
        print('  rows %5d..%5d: %3d steps, mean %.0f us, fit %.0f us' % (lo, hi, m.sum(), us[m].mean(), (A[m] @ [a, b]).mean()))
full = 8192 * 4
print('one batch: %d steps, %d row-steps = %.1f full steps of %d rows; a full-batch walk saves %.1f steps x %.0f us = %.2f ms of %.1f ms (%.1f %%)'
      % (steps, row_steps, row_steps / full, full, steps - row_steps / full, a, (steps - row_steps / full) * a * 1e-3, us.sum() / 3 * 1e-3,
         100 * (steps - row_steps / full) * a / (us.sum() / 3)))
