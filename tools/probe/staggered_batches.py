"""Probe: a walk over several config-2 batches by two host threads on streams of their own, the second starting half a batch
late (so that the small-row tail steps of one batch run beside the large steps of the next), against the same batches one
after the other.  Usage: python tools/probe/staggered_batches.py [batches] [delay_ms]"""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, tst_amd  # noqa: E402,F401
from tst_amd import synthetic  # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
delay = float(sys.argv[2]) * 1e-3 if len(sys.argv) > 2 else 0.088
dev = torch.device('cuda', 0)
model, vocab = bench.build_model(dev, 5.5)
srcs = [synthetic.multi30k_like_sources(8192, bench.VOCAB, seed=2000 + i).to(dev) for i in range(n_batches)]


def decode(i):
    tok, lens, loop = bench.decode_batch(model, vocab, srcs[i], 4, 0.0)
    return tok.cpu(), lens.cpu()


decode(0)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    seq = [decode(i) for i in range(n_batches)]
    torch.cuda.synchronize(); t_seq = time.time() - t0
    out = [None] * n_batches

    def work(k):
        if k:
            time.sleep(delay)
        with torch.cuda.stream(torch.cuda.Stream()):
            for i in range(k, n_batches, 2):
                out[i] = decode(i)

    torch.cuda.synchronize(); t0 = time.time()
    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize(); t_par = time.time() - t0
    same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(seq, out))
    print('%d batches: one after the other %.1f ms | two staggered walkers (delay %.0f ms) %.1f ms (%+.1f %%) same outputs: %s'
          % (n_batches, t_seq * 1e3, delay * 1e3, t_par * 1e3, (t_seq / t_par - 1) * 100, same), flush=True)
