"""Probe: the bench batch (config 2, 8192 sentences) decoded as PARTS concurrent sub-batches, each on a host thread and a HIP
stream of its own (strided split of the length-sorted sentences), against the one-batch decode.  PARTS from argv (default 2)."""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, tst_amd
from tst_amd import synthetic
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda', 0)
model, vocab = bench.build_model(dev, 5.5)
src = synthetic.multi30k_like_sources(8192, bench.VOCAB, seed=1236).to(dev)
model.decoder.event_log = None
subs = [src[i::parts].contiguous() for i in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]
out = [None] * parts


def work(i):
    with torch.cuda.stream(streams[i]):
        tok, lens, loop = bench.decode_batch(model, vocab, subs[i], 4, 0.0)
        out[i] = (tok.cpu(), lens.cpu())


for rep in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    tok, lens, loop = bench.decode_batch(model, vocab, src, 4, 0.0)
    th, lh = tok.cpu(), lens.cpu()
    n1 = int((lh.long() - 1).clamp(min=0).sum())
    t1 = time.time() - t0
    torch.cuda.synchronize(); t0 = time.time()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    t2 = time.time() - t0
    n2 = sum(int((o[1].long() - 1).clamp(min=0).sum()) for o in out)
    same = all(torch.equal(out[i][0][:, :th.shape[1]] if out[i][0].shape[1] >= th.shape[1] else out[i][0], th[i::parts][:, :out[i][0].shape[1]]) for i in range(parts))
    print('one batch %.1f ms (%d tokens) | %d concurrent parts %.1f ms (%d tokens) same tokens: %s' % (t1 * 1e3, n1, parts, t2 * 1e3, n2, same), flush=True)
