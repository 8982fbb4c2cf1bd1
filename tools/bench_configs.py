"""Throughput of the BASELINE configurations other than the one bench.py reports (they are parity-test cases;
this script only measures them for the design notes).  One JSON line per case.

  cfg1: transformer-mt (the bench.py model) at 128 / 256 sentences, greedy (not part of 'all')
  cfg3: GRU decoder (cyclegan sizes: E = H = 1024), 8192 sentences, greedy and beam 4, max_len 100,
        character vocabulary (V = 128) and a 32000-word variant
  cfg4: z-conditioned transformer LM, model_size 512, 6 layers, 8 heads, d_ff 2048, V = 32000, beam 8, bf16
        (and the same in fp32 for comparison)

Usage: python tools/bench_configs.py [cfg1|cfg3|cfg4|all] [--sentences N] [--streams S]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tst_amd
from tst_amd import synthetic


def run(name, make_state, n_sent, repeats=3):
    st = make_state()
    st.decode_packed()                      # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    emitted = steps = rows = 0
    for _ in range(repeats):
        st = make_state()
        tok, lens, _ = st.decode_packed()
        lens_h = lens.cpu()
        emitted += int((lens_h.long() - 1).clamp(min=0).sum())
        steps += st.loop.steps_done
        rows += st.loop.rows_stepped
    torch.cuda.synchronize()
    dt = (time.time() - t0) / repeats
    print(json.dumps({'case': name, 'sentences': n_sent, 'tokens_per_s': emitted / repeats / dt, 'ms_per_batch': dt * 1e3,
                      'mean_emitted_len': emitted / repeats / n_sent, 'decode_steps': steps / repeats,
                      'row_steps': rows / repeats, 'row_steps_per_s': rows / repeats / dt}), flush=True)


def cfg1(n_sent, bias=None):
    """transformer-mt at the reference's own batch sizes, greedy: the launch-bound end of the path."""
    import bench
    dev = torch.device('cuda', 0)
    model, vocab = bench.build_model(dev, 5.5 if bias is None else bias)
    for B in sorted({128, 256, n_sent}):
        src = synthetic.multi30k_like_sources(B, bench.VOCAB, seed=1235).to(dev)

        def make():
            encs, mask = model.encoder(src)
            return tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab,
                                           'max_len': bench.MAX_LEN})
        run('cfg1 transformer-mt V=32000 greedy fp32 max_len=50 (encoder included)', make, B, repeats=5)


def cfg1_streams(n_sent, n_streams, bias=None, repeats=8):
    """The same small batches, `n_streams` of them at a time: every host thread decodes its own batches on a HIP stream of its
    own (independent InferenceStates over one model; the library is safe for one thread per stream).  A batch of this size
    is a chain of dependent small launches that leaves most of the GPU idle, so independent batches overlap almost freely:
    the line reports the aggregate rate and the latency of a batch under that load."""
    import threading
    import bench
    dev = torch.device('cuda', 0)
    model, vocab = bench.build_model(dev, 5.5 if bias is None else bias)
    srcs = [synthetic.multi30k_like_sources(n_sent, bench.VOCAB, seed=1235 + i).to(dev) for i in range(n_streams)]
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    stats = [None] * n_streams

    def work(i, reps):
        torch.cuda.set_device(dev)
        emitted, lat = 0, []
        with torch.cuda.stream(streams[i]):
            for _ in range(reps):
                t0 = time.time()
                encs, mask = model.encoder(srcs[i])
                st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab,
                                             'max_len': bench.MAX_LEN})
                tok, lens, _ = st.decode_packed()
                emitted += int((lens.cpu().long() - 1).clamp(min=0).sum())
                lat.append(time.time() - t0)
        stats[i] = (emitted, lat)

    def round_(reps):
        ths = [threading.Thread(target=work, args=(i, reps)) for i in range(n_streams)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()

    round_(1)                               # warm-up
    t0 = time.time()
    round_(repeats)
    dt = time.time() - t0
    emitted = sum(s_[0] for s_ in stats)
    lat = sorted(x for s_ in stats for x in s_[1])
    print(json.dumps({'case': 'cfg1 transformer-mt V=32000 greedy fp32 max_len=50 (encoder included), concurrent batches',
                      'sentences_per_batch': n_sent, 'streams': n_streams, 'batches': n_streams * repeats,
                      'tokens_per_s': emitted / dt, 'ms_per_batch_amortised': dt * 1e3 / (n_streams * repeats),
                      'batch_latency_ms_median': lat[len(lat) // 2] * 1e3, 'batch_latency_ms_max': lat[-1] * 1e3}), flush=True)


def gru_cell_roofline(model, rows):
    """The GRU cell step alone (x Wih, h Whh and the gate update as one kernel) at `rows` rows, HIP events over 20 launches:
    algorithmic work 2 * rows * 3H * (E + H) FLOP against the dense fp32 matrix peak of MI355X (157.3 TFLOP/s)."""
    from tst_amd import kernels as K
    W = model.prepared()['cells'][0]
    E, H = W['wih'].shape[0], W['whh'].shape[0]
    x = torch.randn((rows, E), device=model.device)
    h = torch.randn((rows, H), device=model.device)
    out = torch.empty_like(h)
    fn = lambda: K.gru_cell_fused(x, h, W['wih'], W['bih'], W['whh'], W['bhh'], out, rows)   # noqa: E731
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tf = 2.0 * rows * 3 * H * (E + H) / us / 1e6
    print(json.dumps({'kernel': 'gru_fused_kernel (GRU cell step, E=%d H=%d)' % (E, H), 'rows': rows, 'avg_launch_us': round(us, 1),
                      'roofline': {'bound': 'mfma', 'achieved': round(tf, 1), 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': round(tf / 157.3, 3)}}),
          flush=True)


def gru_vocab_roofline(model, rows):
    """The output projection H -> V of the GRU decoder at `rows` rows as the decode step runs it for large batches (state
    k-major, both operands by LDS-DMA), HIP events over 10 launches: 2 * rows * H * V FLOP against 157.3 TFLOP/s."""
    from tst_amd import kernels as K
    W = model.prepared()
    H, V = W['wout'].shape[0], model.V
    ht = torch.randn((H, (rows + 127) // 128 * 128), device=model.device)
    out = torch.empty((rows, (V + 3) // 4 * 4), device=model.device)
    fn = lambda: K.gemm_dma(ht, W['wout'], W['bout'], out, rows, n_cols=V, variant=3)   # noqa: E731
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tf = 2.0 * rows * H * V / us / 1e6
    print(json.dumps({'kernel': 'gemm_f32_dma_kernel (GRU output projection, K=%d V=%d)' % (H, V), 'rows': rows, 'avg_launch_us': round(us, 1),
                      'roofline': {'bound': 'mfma', 'achieved': round(tf, 1), 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': round(tf / 157.3, 3)}}),
          flush=True)
    del ht, out


def cfg3(n_sent, bias=None):
    dev = torch.device('cuda', 0)
    for V, eos_bias in ((128, 0.5), (32000, 1.0)):
        eos_bias = eos_bias if bias is None else bias
        vocab = tst_amd.Vocab.synthetic(V, extra=('|',))
        model = tst_amd.RNNDecoder(1024, 1024, vocab, device=dev)
        model.load_state_dict(synthetic.synthetic_state_dict(model, vocab.stoi['|'], eos_bias, 2.0, seed=42))
        z = torch.randn((n_sent, 1024), generator=torch.Generator().manual_seed(1237)).to(dev)
        for beam in (1, 4):
            gru_cell_roofline(model, n_sent * beam)
            if V >= 2048:
                gru_vocab_roofline(model, n_sent * beam)
            run('cfg3 GRU H=1024 V=%d beam=%d fp32 max_len=100' % (V, beam),
                lambda: tst_amd.InferenceState({'model': model, 'inputs': z, 'vocab': vocab, 'max_len': 100, 'eos_token': '|',
                                                'beam_size': beam}), n_sent)


class _Parts:
    """The sentences dealt to `parts` independent InferenceStates that decode at the same time on streams of their own
    (tst_amd.run_on_streams); outputs put back in order, counters summed."""

    class _Loop:
        steps_done = rows_stepped = 0

    def __init__(self, make, z, parts):
        self.parts, self.n = parts, z.shape[0]
        self.states = [make(z[p::parts].contiguous()) for p in range(parts)]
        self.loop = self._Loop()

    def decode_packed(self):
        outs = tst_amd.run_on_streams([st.decode_packed for st in self.states])
        width = max(o[0].shape[1] for o in outs)
        tok = torch.zeros((self.n, width), dtype=outs[0][0].dtype, device=outs[0][0].device)
        lens = torch.empty(self.n, dtype=outs[0][1].dtype, device=tok.device)
        score = torch.empty(self.n, dtype=outs[0][2].dtype, device=tok.device)
        for p, (t, n, sc) in enumerate(outs):
            tok[p::self.parts, :t.shape[1]] = t
            lens[p::self.parts] = n
            score[p::self.parts] = sc
        self.loop.steps_done = max(st.loop.steps_done for st in self.states)
        self.loop.rows_stepped = sum(st.loop.rows_stepped for st in self.states)
        return tok, lens, score


def cfg4(n_sent, bias=None, dtypes=('bf16', 'fp32'), parts=1):
    dev = torch.device('cuda', 0)
    V = 32000
    vocab = tst_amd.Vocab.synthetic(V)
    hp = dict(model_size=512, n_layers=6, n_heads=8, d_ff=2048, pe_max_len=64)
    z = (0.3 * torch.randn((n_sent, 512), generator=torch.Generator().manual_seed(1238))).to(dev)
    for dtype in dtypes:
        lm = tst_amd.TransformerLM(tst_amd.HP(hp), vocab, device=dev, dtype=dtype)
        lm.load_state_dict(synthetic.synthetic_state_dict(lm, 3, 5.0 if bias is None else bias, 2.0, seed=42))
        make = lambda zz: tst_amd.InferenceState({'model': lm, 'inputs': zz, 'vocab': vocab, 'max_len': 50, 'beam_size': 8})   # noqa: E731
        run('cfg4 TransformerLM d=512 L=6 V=32000 beam=8 %s max_len=50%s' % (dtype, '' if parts == 1 else ' (%d concurrent sub-batches)' % parts),
            (lambda: make(z)) if parts == 1 else (lambda: _Parts(make, z, parts)), n_sent, repeats=2)
        del lm
        torch.cuda.empty_cache()


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('which', nargs='?', default='all')
    ap.add_argument('--sentences', type=int, default=8192)
    ap.add_argument('--dtype', default=None, help='cfg4: only this dtype (bf16 or fp32)')
    ap.add_argument('--eos-bias', type=float, default=None, help='override the calibrated EOS bias (calibration runs)')
    ap.add_argument('--streams', type=int, default=0, help='cfg1: that many batches at a time, one host thread and HIP stream each')
    ap.add_argument('--parts', type=int, default=1, help='cfg4: the batch as that many sub-batches decoded at the same time on streams')
    a = ap.parse_args()
    if a.which == 'cfg1' and a.streams > 0:
        cfg1_streams(a.sentences, a.streams, a.eos_bias)
    elif a.which == 'cfg1':
        cfg1(a.sentences, a.eos_bias)
    if a.which in ('cfg3', 'all'):
        cfg3(a.sentences, a.eos_bias)
    if a.which in ('cfg4', 'all'):
        cfg4(a.sentences, a.eos_bias, ('bf16', 'fp32') if a.dtype is None else (a.dtype,), a.parts)
