"""Benchmark of the decode hot path on MI355X.

One "step" = one pass of the path over one batch of synthetic input: encoder pass, beam-search decode
loop and the device-to-host copy of the finished sequences, for BATCH sentences per GPU
(BASELINE.json config 2: transformer-mt base transformer, random init, beam=4, 8k-sentence synthetic
batch, fp32, 32k vocabulary).  With N > 1 GPUs (launched by torch.distributed.run, one rank per GPU) every
rank decodes its own shard of N * BATCH sentences and the outputs are all-gathered over RCCL at the end
of the step ("weak" scaling: the per-GPU work is fixed).

Prints ONE JSON line on rank 0.  ``value`` = decoded tokens (emitted tokens of the returned hypotheses,
BOS excluded) of all ranks per second of the slowest rank.  ``roofline`` describes the dominant kernel,
the vocabulary-projection GEMM on the fp32 matrix cores, timed with HIP events around each of its
launches inside the timed steps.  ``cpu_baseline`` times the CPU restatement of the reference algorithm
(greedy, whole-prefix recompute like /root/reference/src/inference.py:132) on a bounded sample of
BASELINE.json config 1 on the host cores (rank 0, N = 1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HP = dict(d_model=256, dropout=0.1, n_layers=2, pe_max_len=60, n_heads=2, d_ff=1024)
VOCAB = 32000
MAX_LEN = 50
OUT_SCALE = 2.0
FP32_MATRIX_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak of MI355X
HBM_PEAK_GBS = 8000.0            # HBM3E peak of MI355X
TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'r01_vocab_gemm_traffic.json')   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE


def build_model(device, eos_bias):
    import tst_amd
    from tst_amd import synthetic
    vocab = tst_amd.Vocab.synthetic(VOCAB)
    model = tst_amd.Transformer(tst_amd.HP(HP), vocab, vocab, device=device)
    model.load_state_dict(synthetic.synthetic_state_dict(model, vocab.stoi['<eos>'], eos_bias, OUT_SCALE, seed=42))
    return model, vocab


def decode_batch(model, vocab, src, beam, alpha, select_log=None):
    """One pass of the hot path: encoder, decode loop; returns device tensors and the loop (for counters)."""
    import tst_amd
    encs, enc_mask = model.encoder(src)
    st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab,
                                 'max_len': MAX_LEN, 'beam_size': beam, 'length_penalty': alpha})
    st.select_log = select_log      # HIP events around every log-softmax/top-k launch (None: not timed)
    tok, lens, _ = st.decode_packed()   # device tensors; the per-sentence list of the reference API is not built
    return tok, lens, st.loop


def cpu_baseline(eos_bias, sample, min_seconds=10.0):
    """Reference algorithm on the host cores: torch fp32, greedy, whole prefix recomputed every step."""
    from oracle import torch_ref as R
    from tst_amd import synthetic
    threads = max(1, min(os.cpu_count() or 1, 16))
    torch.set_num_threads(threads)
    vocab = R.synthetic_vocab(VOCAB)
    model = R.synthetic_init(R.Transformer(R.HP(HP), vocab, vocab), vocab.stoi['<eos>'], eos_bias, OUT_SCALE, seed=42)
    src = synthetic.multi30k_like_sources(256, VOCAB, seed=1235)[:sample]
    toks = passes = 0
    t0 = time.time()
    with torch.no_grad():
        while passes < 8 and time.time() - t0 < min_seconds:   # whole passes over the sample until ~10 s of CPU work
            encs, enc_mask = model.encoder(src)
            out = R.RefInferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab,
                                       'max_len': MAX_LEN}).inference()
            toks += sum(len(s) - 1 for s in out)
            passes += 1
    dt = time.time() - t0
    return {'value': toks / dt, 'unit': 'tokens/s', 'cores': threads, 'kind': 'port',
            'sample': '%d pass(es) over %d of the 256 sentences of BASELINE config 1 (greedy, max_len=%d, V=%d), %d tokens in '
                      '%.1f s; torch CPU fp32 restatement of src/inference.py with whole-prefix recompute'
                      % (passes, sample, MAX_LEN, VOCAB, toks, dt)}


def cpu_baseline_kv_cached(eos_bias, sample, min_seconds=5.0):
    """The same sentences on the host cores with the ALGORITHMIC change of this build but none of its hardware: the
    canonical-arithmetic C oracle (KV cache, one new token per step; OpenMP over rows).  Reported beside ``cpu_baseline``
    so that the speed-up of the incremental algorithm and that of the GPU are not conflated."""
    from oracle import canon as C
    from oracle import torch_ref as R
    from tst_amd import synthetic
    vocab = R.synthetic_vocab(VOCAB)
    model = R.synthetic_init(R.Transformer(R.HP(HP), vocab, vocab), vocab.stoi['<eos>'], eos_bias, OUT_SCALE, seed=42)
    src = synthetic.multi30k_like_sources(256, VOCAB, seed=1235)[:sample]
    dec = C.CanonTransformerDecoder(model.decoder, HP['n_heads'])
    toks = passes = 0
    t0 = time.time()
    while passes < 8 and time.time() - t0 < min_seconds:
        with torch.no_grad():
            encs, enc_mask = model.encoder(src)
        sess = dec.session(encs.numpy(), enc_mask.numpy(), 1, MAX_LEN)
        seqs = C.greedy_decode(sess, src.shape[0], MAX_LEN, vocab.stoi['<bos>'], vocab.stoi['<eos>'])[0]
        toks += sum(len(s) - 1 for s in seqs)
        passes += 1
    dt = time.time() - t0
    return {'value': toks / dt, 'unit': 'tokens/s', 'cores': int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1)),
            'kind': 'port', 'sample': '%d pass(es) over %d sentences of BASELINE config 1, %d tokens in %.1f s; C oracle with a KV '
                                      'cache (fixed-order fp32), greedy' % (passes, src.shape[0], toks, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=8192, help='sentences per GPU per step')
    ap.add_argument('--beam', type=int, default=4)
    ap.add_argument('--alpha', type=float, default=0.0)
    ap.add_argument('--eos-bias', type=float, default=5.5)
    ap.add_argument('--cpu-sample', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help="'nccl' (RCCL over xGMI); 'gloo' only to rehearse the N > 1 code path")
    ap.add_argument('--share-gpu', action='store_true', help='rehearsal: all ranks use GPU 0 (a one-GPU box)')
    args = ap.parse_args()

    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # before the first HIP call: dmabuf IPC for RCCL
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.backend)

    import importlib
    shard = importlib.import_module('text-style-transfer_amd.shard')
    from tst_amd import synthetic
    model, vocab = build_model(device, args.eos_bias)

    # global batch of world * batch sentences, the same on every rank; each rank decodes its shard
    n_total = world * args.batch
    src_all = synthetic.multi30k_like_sources(n_total, VOCAB, seed=1236)
    src_lens = (src_all != 1).sum(1)
    order, per_rank = shard.shard_plan(src_lens, world)
    mine = shard.my_shard(order, per_rank, rank)
    src = src_all[mine].to(device)          # inputs are resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events around every vocabulary-projection and selection launch of the timed steps, made before the timed region
    from tst_amd.timing import LaunchTimer
    launches = args.steps * (MAX_LEN + 2)
    timing, select_timing = LaunchTimer(launches), LaunchTimer(launches)

    def step(timed):
        model.decoder.event_log = timing if timed else None
        tok, lens, loop = decode_batch(model, vocab, src, args.beam, args.alpha, select_timing if timed else None)
        if world > 1:
            if args.backend != 'nccl':          # rehearsal backend: gathers host tensors
                tok, lens = tok.cpu(), lens.cpu()
            tok, lens = shard.gather_outputs(tok, lens, n_total, order, per_rank)
        tok_h, lens_h = tok.cpu(), lens.cpu()   # final device-to-host copy of the outputs
        return tok_h, lens_h, loop

    for _ in range(args.warmup):
        step(False)
    gc.collect()
    gc.disable()      # no collector pauses of the launching thread inside the timed region
    barrier()
    t0 = time.time()
    emitted = 0
    rows_stepped = 0
    n_steps_loop = 0
    for _ in range(args.steps):
        tok_h, lens_h, loop = step(True)
        emitted += int((lens_h.long() - 1).clamp(min=0).sum())   # all ranks' sentences when gathered
        rows_stepped += loop.rows_stepped
        n_steps_loop += loop.steps_done
    barrier()
    dt = time.time() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        gemm_ms, gemm_flops, gemm_launches = timing.totals()
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        sel_ms, sel_bytes, sel_launches = select_timing.totals()
        sel_gbs = sel_bytes / (sel_ms * 1e-3) / 1e9 if sel_ms > 0 else 0.0
        traffic = None   # HBM bytes per vocabulary-projection launch from the committed PMC passes of this command
        if os.path.exists(TRAFFIC_FILE):
            with open(TRAFFIC_FILE) as f:
                traffic = json.load(f).get('hbm_bytes_per_launch')
        result = {
            'metric': 'decoded tokens/sec (beam=%d, 32k BPE)' % args.beam,
            'value': emitted / dt, 'unit': 'tokens/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE config 2: transformer-mt base (d_model 256, 2 layers, 2 heads, d_ff 1024), random '
                                   'init seed 42, V=32000, %d Multi30k-length synthetic sentences per GPU per step, beam=%d, '
                                   'max_len=%d, fp32; step = encoder pass + decode loop + D2H of outputs' % (args.batch, args.beam, MAX_LEN),
                       'sentences_per_gpu': args.batch, 'beam': args.beam, 'max_len': MAX_LEN, 'vocab': VOCAB,
                       'eos_bias': args.eos_bias, 'mean_emitted_len': emitted / (args.steps * n_total),
                       'decode_steps_per_batch': n_steps_loop / args.steps, 'row_steps_per_batch': rows_stepped / args.steps,
                       'parallelism': 'dp%d (sentence shards, all-gather of outputs)' % world},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': FP32_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / FP32_MATRIX_PEAK_TFLOPS, 'traffic': traffic,
                         'kernel': 'gemm_f32_kernel (vocabulary projection [rows,256]x[256,32000], fp32 MFMA)',
                         'launches': gemm_launches, 'avg_launch_ms': gemm_ms / max(1, gemm_launches),
                         'share_of_step_time': gemm_ms * 1e-3 / dt},
            # second kernel by time, the HBM-bound pass over the [rows, V] logits (algorithmic bytes: every logit once)
            'roofline_select': {'bound': 'hbm', 'achieved': sel_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                'frac': sel_gbs / HBM_PEAK_GBS, 'kernel': 'logsoftmax_topk_kernel (log-sum-exp + top-%d over V=%d)'
                                % (args.beam, VOCAB), 'launches': sel_launches,
                                'avg_launch_ms': sel_ms / max(1, sel_launches), 'share_of_step_time': sel_ms * 1e-3 / dt},
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.eos_bias, args.cpu_sample)
            result['cpu_baseline_kv_cached'] = cpu_baseline_kv_cached(args.eos_bias, args.cpu_sample)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
