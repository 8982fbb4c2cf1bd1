"""Batches whose row count x vocabulary (logits) or row count x positions x width (cache) exceeds 2^31 elements: the decode of a
sentence does not depend on the batch it is in, so the output of one very large batch must equal the outputs of the same
sentences decoded in chunks.  Usage: python tools/probe/large_batch.py [sentences] [beam] [chunk] [tfm|gru|lm_bf16|lm_fp32]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import tst_amd
from tst_amd import synthetic


def decode(model, vocab, src, beam):
    encs, mask = model.encoder(src)
    st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': bench.MAX_LEN,
                                 'beam_size': beam})
    tok, lens, score = st.decode_packed()
    return tok.cpu(), lens.cpu(), score.cpu(), st.loop.rows_stepped


def decode_z(model, vocab, z, beam, max_len, eos):
    st = tst_amd.InferenceState({'model': model, 'inputs': z, 'vocab': vocab, 'max_len': max_len, 'eos_token': eos, 'beam_size': beam})
    tok, lens, score = st.decode_packed()
    return tok.cpu(), lens.cpu(), score.cpu(), st.loop.rows_stepped


def main():
    global decode
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 45000
    beam = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
    kind = sys.argv[4] if len(sys.argv) > 4 else 'tfm'
    dev = torch.device('cuda', 0)
    if kind == 'tfm':
        model, vocab = bench.build_model(dev, 5.5)
        src = synthetic.multi30k_like_sources(n, bench.VOCAB, seed=77).to(dev)
    elif kind == 'gru':                      # cyclegan sizes with a word vocabulary
        vocab = tst_amd.Vocab.synthetic(bench.VOCAB, extra=('|',))
        model = tst_amd.RNNDecoder(1024, 1024, vocab, device=dev)
        model.load_state_dict(synthetic.synthetic_state_dict(model, vocab.stoi['|'], 1.0, 2.0, seed=42))
        src = torch.randn((n, 1024), generator=torch.Generator().manual_seed(5)).to(dev)
        decode = lambda m, v, z, b: decode_z(m, v, z, b, 30, '|')      # noqa: E731
    else:                                    # z-conditioned transformer LM, bf16 or fp32
        vocab = tst_amd.Vocab.synthetic(bench.VOCAB)
        hp = dict(model_size=512, n_layers=6, n_heads=8, d_ff=2048, pe_max_len=64)
        model = tst_amd.TransformerLM(tst_amd.HP(hp), vocab, device=dev, dtype=kind[3:])
        model.load_state_dict(synthetic.synthetic_state_dict(model, 3, 5.0, 2.0, seed=42))
        src = (0.3 * torch.randn((n, 512), generator=torch.Generator().manual_seed(6))).to(dev)
        decode = lambda m, v, z, b: decode_z(m, v, z, b, 50, '<eos>')  # noqa: E731
    t0 = time.time()
    tok, lens, score, rows = decode(model, vocab, src, beam)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(json.dumps({'kind': kind, 'sentences': n, 'beam': beam, 'rows': n * beam, 'logit_elements': n * beam * bench.VOCAB,
                      'cache_elements_per_layer': (0 if kind == 'gru' else n * beam * bench.MAX_LEN * (256 if kind == 'tfm' else 512)), 'seconds': dt, 'row_steps': rows,
                      'peak_GB': torch.cuda.max_memory_allocated() / 1e9}), flush=True)
    bad = 0
    for a in range(0, n, chunk):
        t2, l2, s2, _ = decode(model, vocab, src[a:a + chunk], beam)
        w = min(tok.shape[1], t2.shape[1])
        same = (l2 == lens[a:a + chunk]) & (t2[:, :w] == tok[a:a + chunk, :w]).all(1) & (s2 == score[a:a + chunk])
        bad += int((~same).sum())
        print('chunk at %d: %d sentences differ' % (a, int((~same).sum())), flush=True)
    print('RESULT', 'identical' if bad == 0 else '%d sentences differ' % bad)
    sys.exit(0 if bad == 0 else 1)


main()
