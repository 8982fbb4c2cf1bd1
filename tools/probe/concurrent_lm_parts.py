"""Config 4 (bf16 transformer LM, 8192 sentences, beam 8) as N sub-batches decoded at the same time on streams: wall time per
batch, device allocations and reserved memory per iteration.  Usage: python tools/probe/concurrent_lm_parts.py [parts] [reuse]
reuse = 1: the same streams for every iteration (torch.cuda.Stream objects made once)."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tst_amd
from tst_amd import synthetic

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reuse = len(sys.argv) > 2 and sys.argv[2] == '1'
dev = torch.device('cuda', 0)
vocab = tst_amd.Vocab.synthetic(32000)
hp = dict(model_size=512, n_layers=6, n_heads=8, d_ff=2048, pe_max_len=64)
lm = tst_amd.TransformerLM(tst_amd.HP(hp), vocab, device=dev, dtype='bf16')
lm.load_state_dict(synthetic.synthetic_state_dict(lm, 3, 5.0, 2.0, seed=42))
z = (0.3 * torch.randn((8192, 512), generator=torch.Generator().manual_seed(1238))).to(dev)
zs = [z[p::parts].contiguous() for p in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]


def job(p):
    st = tst_amd.InferenceState({'model': lm, 'inputs': zs[p], 'vocab': vocab, 'max_len': 50, 'beam_size': 8})
    return st.decode_packed()


def on_fixed_streams():
    out = [None] * parts

    def work(i):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[i]):
            out[i] = job(i)
    th = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
    [t.start() for t in th]
    [t.join() for t in th]
    return out


for it in range(5):
    torch.cuda.synchronize()
    t0 = time.time()
    if reuse:
        on_fixed_streams()
    else:
        tst_amd.run_on_streams([(lambda p=p: job(p)) for p in range(parts)])
    torch.cuda.synchronize()
    s = torch.cuda.memory_stats()
    print('iteration %d: %.1f ms, device allocations so far %d, frees %d, reserved %.1f GB, allocated %.1f GB' % (
        it, (time.time() - t0) * 1e3, s['num_device_alloc'], s['num_device_free'], torch.cuda.memory_reserved() / 1e9,
        torch.cuda.memory_allocated() / 1e9), flush=True)
