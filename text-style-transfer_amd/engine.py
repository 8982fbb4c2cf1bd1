"""Device-resident decode loop: greedy and beam search over a step-wise model session.

Mirrors the data flow of ``InferenceState.inference`` (``/root/reference/src/inference.py:154-167``):
one model step per generated token, selection of the next token, retirement of sentences that emitted
EOS (stored BOS-led and EOS-stripped, ``:94-100``), compaction of the batch (``:102-120``) and storing
of the unfinished rows after ``max_len`` steps (``:65-69``) — with all bookkeeping in HIP kernels and one
4-byte device-to-host read per step (the number of sentences still alive).

A *session* provides:
  ``V``               vocabulary size
  ``needs_anc``       whether it reads the ancestor table (incremental KV cache)
  ``step(rows, n_rows, t) -> logits``   fp32 GPU tensor [>= n_rows, >= V], unit inner stride
  ``reorder(row_parent, n_rows, identity)``  called after each step with the old row of every new row
  ``takes_soft``      (optional) ``step`` accepts ``soft=`` token distributions [rows, Vp] in place of ``rows.tok``
  ``rows_per_sentence``  (set by the loop before every step) adjacent rows that belong to one sentence
  ``incremental``     (optional) the session consumes one token per step (KV cache / carried state): a caller-provided
                      prefix is then fed token by token before the loop; otherwise ``step`` reads the prefix from ``rows.hist``
Rows: step 0 has one row per sentence (row a = sentence a, per-sentence state in rows 0..B-1 of buffers sized
for B * beam rows); the first ``reorder`` fans every sentence out to ``beam`` consecutive rows.
"""
import torch

from . import kernels as K


def length_penalty_table(max_len, alpha):
    """lp[n] = ((5 + n) / 6) ** alpha for n emitted tokens, as fp32 (all ones for alpha == 0)."""
    return torch.tensor([((5.0 + n) / 6.0) ** alpha for n in range(max_len + 2)], dtype=torch.float32)


def _check_logits(logits, n_rows, V):
    if not (logits.is_cuda and logits.dim() == 2 and logits.stride(1) == 1 and logits.shape[0] >= n_rows and logits.shape[1] >= V):
        raise ValueError('a model step must return logits [>= %d rows, >= %d columns] on the GPU with unit inner stride, got %s'
                         % (n_rows, V, tuple(logits.shape)))


_LP_CACHE = {}


def _lp_on_device(max_len, alpha, dev):
    """The length-penalty table on the device, kept per (length, alpha, device): a decode call would otherwise pay a
    host-to-device copy for a few hundred bytes that never change."""
    key = (max_len, alpha, str(dev))
    if key not in _LP_CACHE:
        if len(_LP_CACHE) > 64:
            _LP_CACHE.clear()
        _LP_CACHE[key] = length_penalty_table(max_len, alpha).to(dev)
    return _LP_CACHE[key]


LAG_MAX_ROWS = 0 if __import__('os').environ.get('TST_NO_LAG') else 2048   # sentences x beams up to which the loop does not wait for a step's survivor count before the next launch


# Greedy steps of sessions that offer it end in arg-max keys instead of logits (one kernel for the final LayerNorm, the vocabulary
# projection and the arg-max); the lagged loop lets the kernel that counts the survivors store the count into page-locked host
# memory itself instead of queueing a copy behind every step.  Either can be switched off for measurements; same outputs.
_os = __import__('os')
KEYS_ON = not _os.environ.get('TST_NO_ARGMAX_KEYS')
HOST_COUNT_ON = not _os.environ.get('TST_NO_HOST_COUNT')

_PINNED = {}


def _pinned_counts(dev):
    """Two int32 of page-locked host memory per (device, host thread, stream) for the late reads of the survivor counts
    (allocating page-locked memory per decode call would cost more than the reads save).  A lagged loop waits for its last
    copy into the buffer before it returns, so a later loop that gets the same buffer never sees a copy still in flight."""
    key = (str(dev), __import__('threading').get_ident(), torch.cuda.current_stream(dev).cuda_stream)
    if key not in _PINNED:
        if len(_PINNED) > 256:
            _PINNED.clear()
        _PINNED[key] = torch.zeros(2, dtype=torch.int32).pin_memory()
    return _PINNED[key]


class DecodeLoop:
    def __init__(self, session, n_sent, beam, max_len, bos, eos, pad, alpha=0.0, device=None, sample=None, gumbel=None,
                 prefix=None):
        if not (1 <= beam <= 8):
            raise ValueError('beam_size must be in 1..8')
        if beam > session.V:
            raise ValueError('beam_size larger than the vocabulary')
        self.s, self.B, self.k, self.T = session, n_sent, beam, max_len
        self.bos, self.eos = int(bos), int(eos)
        self.sample = sample  # None (arg-max) or (temperature, seed): Gumbel-max sampling, beam 1 only
        if sample is not None and beam != 1:
            raise ValueError('sampling and beam search cannot be combined')
        # with no length penalty a finished top beam is final (see tst_retire_compact): end the sentence there
        self.top_ends = beam > 1 and float(alpha) == 0.0
        dev = device or torch.device('cuda', torch.cuda.current_device())
        self.dev = dev
        R = n_sent * beam
        # A caller-provided prefix [B, t0] of token ids (``active_seqs`` of /root/reference/src/inference.py:37) takes the
        # place of the single BOS: the loop then starts at position pos0 = t0 - 1 and every sequence may grow to
        # t0 + max_len tokens; the prefix counts as pos0 tokens already emitted.
        self.prefix = None if prefix is None else prefix.to(device=dev, dtype=torch.int32).contiguous()
        self.pos0 = 0 if prefix is None else self.prefix.shape[1] - 1
        self.Ttot = self.pos0 + max_len
        if gumbel is not None and prefix is not None:
            raise ValueError('an id prefix and gumbel decoding cannot be combined')
        self.cur = K.RowState(R, self.Ttot, dev, session.needs_anc, pad)
        self.nxt = K.RowState(R, self.Ttot, dev, session.needs_anc, pad)
        self.sel = K.Selection(R, dev)
        # Step 0 runs ONE row per sentence (all beams of a sentence would be identical): row a = sentence a.
        # The first merge fans each sentence out to its `beam` row slots.
        self.cur.hist[:, 0] = self.bos
        self.cur.tok.fill_(self.bos)
        self.cur.sent[:n_sent] = torch.arange(n_sent, dtype=torch.int32, device=dev)
        if self.prefix is not None:
            self.cur.hist[:n_sent, :self.pos0 + 1] = self.prefix
            self.cur.tok[:n_sent] = self.prefix[:, -1]
            self.cur.nemit[:n_sent] = self.pos0
            if self.cur.anc is not None and self.pos0 > 0:   # prefix positions live in the sentence's own row
                self.cur.anc[:n_sent, :self.pos0] = self.cur.sent[:n_sent].unsqueeze(1)
        self.lp = _lp_on_device(self.Ttot, float(alpha), dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.out_tok = torch.full((n_sent, self.Ttot + 1), pad, **i32)
        self.out_score = torch.zeros(n_sent, dtype=torch.float32, device=dev)
        shapes = [(n_sent,), (2 * n_sent,), (R,), (1,), (R,), (1,)]      # one allocation for the small int32 arrays
        arena = torch.zeros(K._arena_size(shapes), **i32)
        self.out_len, self.scratch, self.row_parent, self.n_alive, self.am, self.n_alive_b = K._carve(arena, shapes)
        # Small batches are bound by the chain of dependent launches, and a blocking read of the survivor count after every
        # step adds a round trip to it.  There the loop launches step t + 1 with the count it read after step t - 1 (an upper
        # bound: counts only fall), the kernels take the true count from device memory, and the host reads counts one step
        # late from pinned memory.  Large batches keep the exact count (extra rows would cost real time there).
        self.lagged = (R <= LAG_MAX_ROWS and gumbel is None and getattr(session, 'lag_ok', False))
        self.top_val = self.top_idx = self.lse = None
        if beam > 1:
            self.top_val = torch.zeros((R, beam), dtype=torch.float32, device=dev)
            self.top_idx = torch.zeros((R, beam), **i32)
            self.lse = torch.zeros(R, dtype=torch.float32, device=dev)
        self.tail = None   # kernels.SelectPlan, made at the first step (it needs the session's logits buffer)
        self.keys = None   # arg-max keys of greedy steps (see _keys_for)
        if (KEYS_ON and beam == 1 and sample is None and gumbel is None and getattr(session, 'keys_ok', None) is not None):
            self.keys = torch.zeros((R, K.ARGMAX_SLOTS), dtype=torch.int64, device=dev)
        # Soft decoding (``gumbel=True`` of /root/reference/src/inference.py:104-105,136-138): the next input of a row
        # is the distribution softmax((softmax(logits) + G) / temperature) instead of a token id; the token that
        # decides about EOS is still the arg-max / sample of the raw logits.  gumbel = (temperature, prefix [B, V],
        # draw(n_rows, V) -> uniform numbers on the device).  dist_seq keeps every sentence's distributions.
        self.gumbel = gumbel
        if gumbel is not None:
            if beam != 1:
                raise ValueError('gumbel decoding and beam search cannot be combined')
            if not getattr(session, 'takes_soft', False):
                raise NotImplementedError('this model cannot be stepped on token distributions')
            V = session.V
            self.Vp = (V + 3) // 4 * 4
            f32 = dict(dtype=torch.float32, device=dev)
            self.dist_seq = torch.zeros((n_sent, max_len + 1, V), **f32)
            self.dist_in = torch.zeros((n_sent, self.Vp), **f32)     # input of the coming step, one row per live row
            self.dist_out = torch.zeros((n_sent, self.Vp), **f32)    # output of the step, before compaction
            self.dist_in[:, :V] = gumbel[1]
            self.dist_seq[:, 0] = gumbel[1]
        self.steps_done = 0
        self.rows_stepped = 0
        self.select_log = None   # set to a timing.LaunchTimer to time every log-softmax/top-k launch (work = bytes)

    def _keys_for(self, R, t, keep_logits):
        """The keys buffer when step t (of R rows) may end in arg-max keys instead of logits, else None."""
        if self.keys is None or self.select_log is not None or t < keep_logits or not self.s.keys_ok(R):
            return None
        return self.keys

    def run(self, keep_logits=0):
        """Runs the loop.  Returns (out_tok [B, pos0 + T + 1], out_len [B], out_score [B]) on the device; with
        ``keep_logits`` > 0 also records (logits, sentence ids of the rows) of the first steps in
        ``self.kept`` (a debugging aid for the fp32 tolerance checks)."""
        s, k, T = self.s, self.k, self.T
        n_sent = self.B
        pos0, Ttot = self.pos0, self.Ttot
        self.kept = []
        if pos0 > 0 and getattr(s, 'incremental', False):   # feed the prefix, all but its last token, one token per step
            for p in range(pos0):
                self.cur.tok[:n_sent] = self.prefix[:, p]
                s.step(self.cur, n_sent, p)
            self.cur.tok[:n_sent] = self.prefix[:, -1]
        if self.lagged and self.select_log is None:
            return self._run_lagged(keep_logits)
        for t in range(T):
            k_in = 1 if t == 0 else k      # rows per sentence going into this step
            R = n_sent * k_in
            s.rows_per_sentence = k_in     # sessions that share work among the beams of a sentence read this
            keys = self._keys_for(R, t, keep_logits)
            if keys is not None:
                logits = s.step(self.cur, R, pos0 + t, keys=keys)
            elif self.gumbel is None:
                logits = s.step(self.cur, R, pos0 + t)
            else:
                logits = s.step(self.cur, R, t, soft=self.dist_in)
            if t < keep_logits:
                self.kept.append((logits[:R, :s.V].clone(), self.cur.sent[:R].clone(), self.cur.fin[:R].clone()))
            if self.gumbel is None and self.select_log is None:
                # selection, retirement and compaction: one native call
                if self.tail is None:
                    _check_logits(logits, R, s.V)
                    self.tail = K.SelectPlan(k=k, eos=self.eos, V=s.V, logits=logits, am=self.am, top_val=self.top_val,
                                             top_idx=self.top_idx, lse=self.lse, sel=self.sel, lp=self.lp, T=Ttot,
                                             top_ends=self.top_ends, sample=self.sample, out_tok=self.out_tok, out_len=self.out_len,
                                             out_score=self.out_score, scratch=self.scratch, row_parent=self.row_parent,
                                             n_alive=self.n_alive, stats=getattr(s, 'stats', None))
                elif logits.data_ptr() != self.tail.logits_ptr:
                    _check_logits(logits, R, s.V)
                    self.tail.set_logits(logits)
                self.tail.set_keys(keys)
                self.tail.run(self.cur, self.nxt, n_sent, k_in, pos0 + t, t, t == T - 1)
            else:
                if k == 1:
                    if self.sample is None:
                        K.argmax_rows(logits, R, s.V, self.am)
                    else:
                        K.sample_rows(logits, R, s.V, self.sample[0], self.sample[1], t, self.cur.sent, self.am)
                    K.greedy_select(self.am, R, self.eos, self.sel, self.cur)
                    if self.gumbel is not None:
                        u = self.gumbel[2](R, s.V)
                        K.gumbel_softmax(logits, R, s.V, u, self.gumbel[0], self.dist_out)
                        self.dist_seq[:, t + 1].index_copy_(0, self.cur.sent[:R].long(), self.dist_out[:R, :s.V])
                else:   # measurement aid: HIP events on the launch stream around the selection pass
                    self.select_log.begin()
                    K.logsoftmax_topk(logits, R, s.V, k, self.top_val, self.top_idx, self.lse)
                    self.select_log.end(float(R) * s.V * logits.element_size())
                    K.beam_merge(self.cur, self.top_val, self.top_idx, self.lse, n_sent, k, k_in, self.eos, self.sel)
                K.retire_compact(self.cur, self.nxt, self.sel, self.lp, n_sent, k, pos0 + t, Ttot, self.eos, t == T - 1, self.top_ends,
                                 self.out_tok, self.out_len, self.out_score, self.scratch, self.row_parent, self.n_alive)
            self.cur, self.nxt = self.nxt, self.cur
            self.steps_done += 1
            self.rows_stepped += R
            alive = int(self.n_alive.item())  # the one host read of the step
            if alive == 0:
                break
            s.reorder(self.row_parent, alive * k, identity=(k == 1 and alive == n_sent))
            if self.gumbel is not None:
                K.gather_rows(self.dist_out, self.row_parent, self.dist_in, alive)
            n_sent = alive
        return self.out_tok, self.out_len, self.out_score

    def _run_lagged(self, keep_logits):
        """The loop without a blocking read per step (see ``self.lagged``): same kernels, same results; a finished batch is
        noticed one step late (that step runs on zero sentences)."""
        s, k, T = self.s, self.k, self.T
        pos0, Ttot = self.pos0, self.Ttot
        counts = (self.n_alive, self.n_alive_b)                     # device: survivors of step t in counts[t % 2]
        host = _pinned_counts(self.dev)
        events = (torch.cuda.Event(), torch.cuda.Event())
        bound = self.B                                              # upper bound of the sentences entering the coming step
        last = None
        for t in range(T):
            k_in = 1 if t == 0 else k
            R = bound * k_in
            s.rows_per_sentence = k_in
            keys = self._keys_for(R, t, keep_logits)
            logits = s.step(self.cur, R, pos0 + t) if keys is None else s.step(self.cur, R, pos0 + t, keys=keys)
            if t < keep_logits:
                self.kept.append((logits[:R, :s.V].clone(), self.cur.sent[:R].clone(), self.cur.fin[:R].clone()))
            if self.tail is None:
                _check_logits(logits, R, s.V)
                self.tail = K.SelectPlan(k=k, eos=self.eos, V=s.V, logits=logits, am=self.am, top_val=self.top_val,
                                         top_idx=self.top_idx, lse=self.lse, sel=self.sel, lp=self.lp, T=Ttot,
                                         top_ends=self.top_ends, sample=self.sample, out_tok=self.out_tok, out_len=self.out_len,
                                         out_score=self.out_score, scratch=self.scratch, row_parent=self.row_parent,
                                         n_alive=self.n_alive, stats=getattr(s, 'stats', None))
            elif logits.data_ptr() != self.tail.logits_ptr:
                _check_logits(logits, R, s.V)
                self.tail.set_logits(logits)
            self.tail.set_counts(counts[t % 2], counts[(t - 1) % 2] if t > 0 else None)
            self.tail.set_keys(keys)
            self.tail.set_count_host(host.data_ptr() + 4 * (t % 2) if HOST_COUNT_ON else None)
            self.tail.run(self.cur, self.nxt, bound, k_in, pos0 + t, t, t == T - 1)
            if not HOST_COUNT_ON:
                host[t % 2:t % 2 + 1].copy_(counts[t % 2], non_blocking=True)
            events[t % 2].record()
            last = events[t % 2]
            self.cur, self.nxt = self.nxt, self.cur
            self.steps_done += 1
            self.rows_stepped += R
            if t == 0:                                  # every sentence entered step 0: the bound is exact
                s.reorder(self.row_parent, bound * k, identity=False)
                continue
            events[(t - 1) % 2].synchronize()           # the count after step t - 1: that step ended before step t began
            entering = int(host[(t - 1) % 2])           # = sentences that entered step t >= those that enter step t + 1
            if keep_logits and entering < bound:        # the kept rows of step t beyond the true count were stale
                lg, sent, fin = self.kept[-1] if t < keep_logits else (None, None, None)
                if lg is not None:
                    self.kept[-1] = (lg[:entering * k_in], sent[:entering * k_in], fin[:entering * k_in])
            if entering == 0:                           # step t ran on no sentence at all
                self.steps_done -= 1
                self.rows_stepped -= R
                if t < keep_logits:
                    self.kept.pop()
                break
            s.reorder(self.row_parent, entering * k, identity=False)
            bound = entering
        if last is not None:
            last.synchronize()      # the last copy into the pinned buffer has landed: the buffer may be handed to another loop
        return self.out_tok, self.out_len, self.out_score
