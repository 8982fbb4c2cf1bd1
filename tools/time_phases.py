"""Wall-clock split of one bench step (encoder / session setup / decode loop / output copy)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, tst_amd
from tst_amd import synthetic
model, vocab = bench.build_model(torch.device('cuda', 0), 6.0)
src = synthetic.multi30k_like_sources(8192, bench.VOCAB, seed=1236).cuda()
def sync(): torch.cuda.synchronize(); return time.time()
for it in range(3):
    t0 = sync()
    encs, mask = model.encoder(src)
    t1 = sync()
    st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 50, 'beam_size': 4})
    sess = st._session()
    t2 = sync()
    from tst_amd.engine import DecodeLoop
    loop = DecodeLoop(sess, 8192, 4, 50, 2, 3, 1, 0.0, torch.device('cuda', 0))
    t3 = sync()
    loop.run()
    t4 = sync()
    a, b = loop.out_tok.cpu(), loop.out_len.cpu()
    t5 = sync()
    lst = [a[i, :n] for i, n in enumerate(b.tolist())]
    t6 = time.time()
    print('encoder %.1f ms, session %.1f ms, loop init %.1f ms, loop %.1f ms, D2H %.1f ms, list of tensors %.1f ms' %
          tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)))
