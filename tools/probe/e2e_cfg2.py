"""Probe: wall time of the bench batch (config 2) without per-kernel timing events; DecodeLoop class attributes can be
overridden from the command line, e.g.  PIPE_PARTS=2 PIPE_MIN=4096."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, tst_amd
from tst_amd import synthetic
from tst_amd.engine import DecodeLoop
for a in sys.argv[1:]:
    k, v = a.split('=')
    setattr(DecodeLoop, k, int(v))
dev = torch.device('cuda', 0)
model, vocab = bench.build_model(dev, 5.5)
src = synthetic.multi30k_like_sources(8192, bench.VOCAB, seed=1236).to(dev)
model.decoder.event_log = None
for rep in range(int(os.environ.get("REPS", "4"))):
    torch.cuda.synchronize(); t0 = time.time()
    tok, lens, loop = bench.decode_batch(model, vocab, src, 4, 0.0)
    n = int((lens.cpu().long() - 1).clamp(min=0).sum())
    dt = time.time() - t0
    print('%s  %.1f ms  %.0f tokens/s  checksum %d' % (' '.join(sys.argv[1:]), dt * 1e3, n / dt, int(tok.long().sum())), flush=True)
