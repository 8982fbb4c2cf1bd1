"""Benchmark of the decode hot path on MI355X.

``python bench.py --gpus N --steps K --warmup W``.  With N > 1 and no rank environment this process makes no GPU call:
it starts ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` (fresh rank processes, one per GPU, RCCL),
lets rank 0 print the JSON line and exits with the children's status.  Started by torch.distributed.run itself (WORLD_SIZE
set) it is a rank.

``--config 2`` (default; BASELINE.json config 2, the configuration the metric is quoted on): one "step" = one pass of the
path over one batch of synthetic input: encoder pass, beam-search decode loop and the device-to-host copy of the finished
sequences, for BATCH sentences per GPU (transformer-mt base transformer, random init, beam=4, 8k-sentence synthetic batch,
fp32, 32k vocabulary).  With N > 1 every rank decodes its own shard of N * BATCH sentences and the outputs are all-gathered
over RCCL at the end of the step ("weak" scaling: the per-GPU work is fixed).

``--config 5`` (BASELINE.json config 5): one step = a fixed corpus (``--sentences``, default 1 000 000) decoded once: dealt
to the N ranks, every rank walks its shard in on-GPU batches of BATCH sentences, one all-gather of the outputs, un-permute,
device-to-host copy ("strong" scaling: the total work is fixed).  ``output_checksum`` is the same for every N.

Prints ONE JSON line on rank 0.  ``value`` = decoded tokens (emitted tokens of the returned hypotheses, BOS excluded) of all
ranks per second of the slowest rank, over K steps of the path exactly as a caller gets it (no instrumentation).
``roofline`` describes the dominant kernel, the vocabulary-projection GEMM on the fp32 matrix cores; it comes from a SEPARATE
pass after the timed region, in the same run, with HIP events around each of its launches on the launch stream (that pass
splits a decode step into more native calls, which is why it is not the pass that is timed for ``value``).
``cpu_baseline`` times the CPU restatement of the reference algorithm (greedy, whole-prefix recompute like
/root/reference/src/inference.py:132) on a bounded sample of BASELINE.json config 1 on the host cores, and
``peaks_measured`` holds what this GPU reaches on a copy loop and two MFMA micro-loops (tools/peaks.hip), beside the vendor
peaks the roofline fractions are priced against (rank 0, N = 1 only).
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HP = dict(d_model=256, dropout=0.1, n_layers=2, pe_max_len=60, n_heads=2, d_ff=1024)
VOCAB = 32000
MAX_LEN = 50
OUT_SCALE = 2.0
FP32_MATRIX_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak of MI355X
BF16_MATRIX_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0            # HBM3E peak of MI355X
TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'r03_vocab_gemm_traffic.json')   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
PEAKS_BIN = os.path.join(ROOT, 'tools', 'peaks')


def launch_ranks(n, argv):
    """Parent of an N-GPU run: starts the ranks as fresh processes and relays their exit status.  Makes no GPU call."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def build_model(device, eos_bias):
    import tst_amd
    from tst_amd import synthetic
    vocab = tst_amd.Vocab.synthetic(VOCAB)
    model = tst_amd.Transformer(tst_amd.HP(HP), vocab, vocab, device=device)
    model.load_state_dict(synthetic.synthetic_state_dict(model, vocab.stoi['<eos>'], eos_bias, OUT_SCALE, seed=42))
    return model, vocab


class _Counts:
    """steps_done / rows_stepped of the parts of a batch that was decoded as concurrent sub-batches"""
    def __init__(self, loops):
        self.steps_done = max(lp.steps_done for lp in loops)
        self.rows_stepped = sum(lp.rows_stepped for lp in loops)


def decode_batch(model, vocab, src, beam, alpha, select_log=None, parts=1):
    """One pass of the hot path: encoder, decode loop; returns device tensors and the loop (for counters).
    parts > 1: the batch's sentences are dealt to that many sub-batches (src[p::parts]: the same length profile each), which are
    decoded at the same time on streams of their own (tst_amd.run_on_streams) and put back in order."""
    import torch
    import tst_amd
    if parts > 1 and src.shape[0] >= 2 * parts:
        outs = tst_amd.run_on_streams([(lambda p=p: decode_batch(model, vocab, src[p::parts].contiguous(), beam, alpha)) for p in range(parts)])
        tok = torch.empty((src.shape[0], outs[0][0].shape[1]), dtype=outs[0][0].dtype, device=src.device)
        lens = torch.empty(src.shape[0], dtype=outs[0][1].dtype, device=src.device)
        for p, (t, n, _) in enumerate(outs):
            tok[p::parts] = t
            lens[p::parts] = n
        return tok, lens, _Counts([o[2] for o in outs])
    encs, enc_mask = model.encoder(src)
    st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab,
                                 'max_len': MAX_LEN, 'beam_size': beam, 'length_penalty': alpha})
    st.select_log = select_log      # HIP events around every log-softmax/top-k launch (None: not timed)
    tok, lens, _ = st.decode_packed()   # device tensors; the per-sentence list of the reference API is not built
    return tok, lens, st.loop


def host_info():
    model = None
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = None
    return {'cpu_model': model, 'os_cpu_count': os.cpu_count(), 'affinity_cpus': affinity,
            'omp_num_threads_env': os.environ.get('OMP_NUM_THREADS')}


def cpu_threads():
    """Threads for the CPU baseline: every CPU this process may run on, within the share the environment grants
    (OMP_NUM_THREADS when it is set: a GPU box hands one job a fixed share of a much larger machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    cap = os.environ.get('OMP_NUM_THREADS')
    if cap and cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def cpu_baseline(eos_bias, sample, min_seconds=10.0):
    """Reference algorithm on the host cores: torch fp32, greedy, whole prefix recomputed every step."""
    import torch
    from oracle import torch_ref as R
    from tst_amd import synthetic
    threads = cpu_threads()
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
    res = {'value': toks / dt, 'unit': 'tokens/s', 'cores': threads, 'threads': torch.get_num_threads(), 'kind': 'port',
           'sample': '%d pass(es) over %d of the 256 sentences of BASELINE config 1 (greedy, max_len=%d, V=%d), %d tokens in '
                     '%.1f s; torch CPU fp32 restatement of src/inference.py with whole-prefix recompute'
                     % (passes, sample, MAX_LEN, VOCAB, toks, dt)}
    res.update(host_info())
    return res


def cpu_baseline_kv_cached(eos_bias, sample, min_seconds=5.0):
    """The same sentences on the host cores with the ALGORITHMIC change of this build but none of its hardware: the
    canonical-arithmetic C oracle (KV cache, one new token per step; OpenMP over rows).  Reported beside ``cpu_baseline``
    so that the speed-up of the incremental algorithm and that of the GPU are not conflated."""
    import torch
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
    return {'value': toks / dt, 'unit': 'tokens/s', 'cores': cpu_threads(),
            'kind': 'port', 'sample': '%d pass(es) over %d sentences of BASELINE config 1, %d tokens in %.1f s; C oracle with a KV '
                                      'cache (fixed-order fp32), greedy' % (passes, src.shape[0], toks, dt)}


def peaks_measured():
    """Runs the peak probe (its own process) and returns its JSON, or None when the binary is not built."""
    if not os.path.exists(PEAKS_BIN):
        return None
    try:
        out = subprocess.run([PEAKS_BIN], capture_output=True, text=True, timeout=120)
        if out.returncode != 0:
            return {'error': out.stderr.strip()[-300:]}
        res = json.loads(out.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, subprocess.TimeoutExpired) as e:
        return {'error': str(e)}
    res['vendor'] = {'hbm_GBps': HBM_PEAK_GBS, 'mfma_f32_TFLOPs': FP32_MATRIX_PEAK_TFLOPS, 'mfma_bf16_TFLOPs': BF16_MATRIX_PEAK_TFLOPS}
    return res


def checksum(tok, lens):
    """Order-sensitive 61-bit checksum of the outputs (host tensors): equal outputs <=> equal value, for every N."""
    import torch
    t = tok.long()
    cols = torch.arange(t.shape[1], dtype=torch.int64)
    valid = cols[None, :] < lens.long()[:, None]
    rows = torch.arange(t.shape[0], dtype=torch.int64)[:, None]
    mix = ((t + 1) * (cols[None, :] * 1000003 + 7) + rows * 8191) % 2305843009213693951
    return int((mix * valid).sum() % 2305843009213693951)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default 5; 1 for --config 5)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed steps before (default 2; 1 for --config 5)')
    ap.add_argument('--config', type=int, default=2, choices=(2, 5), help='BASELINE.json config: 2 = one 8k-sentence batch per GPU '
                    'per step (weak scaling, the headline); 5 = a fixed corpus sharded over the GPUs (strong scaling)')
    ap.add_argument('--sentences', type=int, default=1000000, help='--config 5: sentences in the corpus')
    ap.add_argument('--batch', type=int, default=8192, help='sentences per GPU per on-GPU batch')
    ap.add_argument('--beam', type=int, default=4)
    ap.add_argument('--sub-batches', type=int, default=1, help='decode every on-GPU batch as that many concurrent sub-batches on '
                    'streams of their own (same sentences, same outputs)')
    ap.add_argument('--alpha', type=float, default=0.0)
    ap.add_argument('--eos-bias', type=float, default=5.5)
    ap.add_argument('--cpu-sample', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true', help='skip the separate event-timed pass')
    ap.add_argument('--dump', default=None, help='rank 0 saves the outputs of the last step (torch.save of (tok, lens)) here')
    ap.add_argument('--backend', default='nccl', help="'nccl' (RCCL over xGMI); 'gloo' only to rehearse the N > 1 code path")
    ap.add_argument('--share-gpu', action='store_true', help='rehearsal: all ranks use GPU 0 (a one-GPU box)')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 1 if args.config == 5 else 5
    if args.warmup is None:
        args.warmup = 1 if args.config == 5 else 2   # the first batches on a fresh box run 2-5 % slow (clocks, first touches)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # before the first HIP call: dmabuf IPC for RCCL
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if world > 1:
        # N ranks share this host: each takes its share of the cores the job may use (set-up would otherwise run N x all threads)
        share = max(1, cpu_threads() // world)
        os.environ['OMP_NUM_THREADS'] = str(share)
    import torch
    if world > 1:
        torch.set_num_threads(share)
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
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

    # the sentences of the job: every rank derives the same plan from the sentence LENGTHS of the corpus (cheap) and then
    # builds only its own shard of it (same seed, same random stream: the rows other ranks own are dropped chunk by chunk);
    # the shard is resident in HBM before the timed region
    n_total = args.sentences if args.config == 5 else world * args.batch
    corpus_seed = 1236 if args.config == 2 else 1240
    src_lens = torch.from_numpy(synthetic.multi30k_like_lengths(n_total, corpus_seed) + 2)     # + BOS and EOS
    order, per_rank = shard.shard_plan(src_lens, world)
    mine = shard.my_shard(order, world, rank)
    mine_sorted, back = torch.sort(mine)
    src = synthetic.multi30k_like_sources(n_total, VOCAB, seed=corpus_seed, rows=mine_sorted.numpy())
    src = src[torch.argsort(back)].to(device)      # walking order: the rank's sentences by length
    my_lens = src_lens[mine]            # host: the walk trims every batch to its longest sentence

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    counters = {'rows': 0, 'loop_steps': 0, 'batches': 0, 'gather_s': 0.0}

    def decode_fn(batch):
        tok, lens, loop = decode_batch(model, vocab, batch, args.beam, args.alpha, parts=args.sub_batches)
        counters['rows'] += loop.rows_stepped
        counters['loop_steps'] += loop.steps_done
        counters['batches'] += 1
        return tok, lens

    def step():
        """The path as a caller gets it: no event log, selection tail as one native call."""
        model.decoder.event_log = None
        if args.config == 5:
            tok, lens = shard.decode_shard(decode_fn, src, None, args.batch, my_lens)
        else:
            tok, lens = decode_fn(src)
        if world > 1:
            # the decode is complete on this rank before its outputs can be gathered; the two waits around the gather only
            # put a number on it (time from this rank's last kernel to the gathered, un-permuted outputs: the collective
            # itself plus the wait for the slowest rank)
            torch.cuda.synchronize()
            tg = time.time()
            if args.backend != 'nccl':          # rehearsal backend: gathers host tensors
                tok, lens = tok.cpu(), lens.cpu()
            tok, lens = shard.gather_outputs(tok, lens, n_total, order, per_rank)
            torch.cuda.synchronize()
            counters['gather_s'] += time.time() - tg
        else:
            tok, lens = shard.unpermute(tok, lens, order)
        return tok.cpu(), lens.cpu()            # final device-to-host copy of the outputs

    for _ in range(args.warmup):
        step()
    for k in counters:
        counters[k] = type(counters[k])(0)
    gc.collect()
    gc.disable()      # no collector pauses of the launching thread inside the timed region
    barrier()
    t0 = time.time()
    emitted = 0
    for _ in range(args.steps):
        tok_h, lens_h = step()
        emitted += int((lens_h.long() - 1).clamp(min=0).sum())   # all ranks' sentences when gathered
    barrier()
    dt = time.time() - t0
    gc.enable()
    rank_ms = gather_ms = None
    if world > 1:
        # every rank's own clock over the timed region and its time in the gather: `value` uses the slowest rank; the
        # per-rank list shows a straggler, a long gather on the fast ranks is the wait for it
        t = torch.tensor([dt, counters['gather_s']], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
        every = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(e[0]) / args.steps * 1e3 for e in every]
        gather_ms = [float(e[1]) / args.steps * 1e3 for e in every]
        dt = max(float(e[0]) for e in every)

    # Separate pass for the roofline figures: HIP events (made before the pass) on the launch stream around every
    # vocabulary-projection and selection launch of one batch.  Not part of `value`.
    gemm = sel = None
    if rank == 0 and not args.no_roofline:
        from tst_amd.timing import LaunchTimer
        launches = MAX_LEN + 2
        timing, select_timing = LaunchTimer(launches), LaunchTimer(launches)
        model.decoder.event_log = timing
        decode_batch(model, vocab, src[:args.batch, :int(my_lens[:args.batch].max())], args.beam, args.alpha, select_timing)
        model.decoder.event_log = None
        torch.cuda.synchronize()
        gemm, sel = timing.totals(), select_timing.totals()

    if rank == 0:
        if args.dump:
            torch.save((tok_h, lens_h), args.dump)
        n_batches = max(1, counters['batches'])
        workload = ('BASELINE config 2: transformer-mt base (d_model 256, 2 layers, 2 heads, d_ff 1024), random init seed 42, '
                    'V=%d, %d Multi30k-length synthetic sentences per GPU per step, beam=%d, max_len=%d, fp32; step = encoder pass + '
                    'decode loop + D2H of outputs' % (VOCAB, args.batch, args.beam, MAX_LEN))
        if args.config == 5:
            workload = ('BASELINE config 5: the config-2 model, a corpus of %d synthetic sentences dealt to %d rank(s), walked in '
                        'on-GPU batches of %d, beam=%d, max_len=%d, fp32; step = the whole corpus: encoder passes + decode loops + '
                        'all-gather + D2H of outputs' % (n_total, world, args.batch, args.beam, MAX_LEN))
        result = {
            'metric': 'decoded tokens/sec (beam=%d, 32k BPE)' % args.beam,
            'value': emitted / dt, 'unit': 'tokens/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if args.config == 5 else 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload, 'baseline_config': args.config, 'sentences_per_step': n_total,
                       'sentences_per_gpu_batch': args.batch, 'sub_batches': args.sub_batches, 'beam': args.beam, 'max_len': MAX_LEN,
                       'vocab': VOCAB,
                       'eos_bias': args.eos_bias, 'mean_emitted_len': emitted / (args.steps * n_total),
                       'mean_emitted_len_note': 'eos_bias 5.5 gives ~23 tokens per sentence (~15 was the planning figure); longer '
                                                'sentences mean more decode steps per batch, not fewer',
                       'decode_steps_per_batch': counters['loop_steps'] / n_batches,
                       'row_steps_per_batch': counters['rows'] / n_batches,
                       'batches_per_step_rank0': counters['batches'] / args.steps,
                       'ranks': world, 'backend': (args.backend if world > 1 else None),
                       'ranks_seen_by_backend': (dist.get_world_size() if world > 1 else 1),
                       'parallelism': 'dp%d (length-sorted sentences dealt round-robin, no in-loop collective, one all-gather of '
                                      'outputs)' % world},
            'output_checksum': checksum(tok_h, lens_h),
            'rank_ms': rank_ms, 'gather_ms': gather_ms,     # per rank, per step (None for one rank)
            'host_threads_per_rank': torch.get_num_threads(),
            'timed_path': 'un-instrumented product path (tst_tfm_step_run + tst_select_step_run per decode step)',
        }
        if gemm is not None:
            gemm_ms, gemm_flops, gemm_launches = gemm
            achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
            sel_ms, sel_bytes, sel_launches = sel
            sel_gbs = sel_bytes / (sel_ms * 1e-3) / 1e9 if sel_ms > 0 else 0.0
            traffic = None   # HBM bytes per vocabulary-projection launch from the committed PMC passes of this command
            if os.path.exists(TRAFFIC_FILE):
                with open(TRAFFIC_FILE) as f:
                    traffic = json.load(f).get('hbm_bytes_per_launch')
            batch_ms = dt / args.steps * 1e3 / (counters['batches'] / args.steps)
            result['roofline'] = {
                'bound': 'mfma', 'achieved': achieved, 'peak': FP32_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / FP32_MATRIX_PEAK_TFLOPS, 'traffic': traffic,
                'traffic_source': 'rocprofv3 --pmc passes of this command, committed under profiles/ (not collected in this run)',
                'kernel': 'gemm_f32_wreg_kernel (vocabulary projection [rows,256]x[256,32000], fp32 MFMA, the weights held in registers, '
                          'A tiles through LDS; gemm_f32_kernel for launches under 1024 rows)',
                'launches': gemm_launches, 'avg_launch_ms': gemm_ms / max(1, gemm_launches),
                'share_of_batch_time': gemm_ms / batch_ms,
                'pass': 'separate event-timed pass over one batch after the timed region (HIP events on the launch stream)'}
            # second kernel by time, the HBM-bound pass over the [rows, V] logits (algorithmic bytes: every logit once)
            result['roofline_select'] = {
                'bound': 'hbm', 'achieved': sel_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': sel_gbs / HBM_PEAK_GBS,
                'kernel': 'logsoftmax_topk_kernel (log-sum-exp + top-%d over V=%d)' % (args.beam, VOCAB),
                'launches': sel_launches, 'avg_launch_ms': sel_ms / max(1, sel_launches), 'share_of_batch_time': sel_ms / batch_ms}
        if world == 1:
            torch.cuda.empty_cache()
            pk = peaks_measured()
            if pk is not None:
                result['peaks_measured'] = pk
            if not args.no_cpu_baseline:
                result['cpu_baseline'] = cpu_baseline(args.eos_bias, args.cpu_sample)
                result['cpu_baseline_kv_cached'] = cpu_baseline_kv_cached(args.eos_bias, args.cpu_sample)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
