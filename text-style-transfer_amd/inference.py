"""``InferenceState`` / ``simple_inference`` with the call surface of ``/root/reference/src/inference.py``.

Same constructor dict and defaults (``:12-45``), same checks (``:50-58``), same results: a list, in
original batch order, of 1-D id tensors that start with BOS and do not contain EOS, cut after
``max_len`` generated tokens (``:94-100``, ``:65-69``, ``:154-167``).  What differs is where the work
happens: token selection, EOS retirement, batch compaction and the hypothesis bookkeeping run in HIP
kernels on device-resident state (``engine.DecodeLoop``), and models of this package are stepped
incrementally (KV cache / carried GRU state) instead of being re-run on the whole prefix (``:132``).

Additive keys (not in the reference): ``dtype`` / ``return_logits`` (see the constructor; a ``dtype`` that differs from the
model's switches the model for this call only), ``beam_size`` (1 = the reference's greedy arg-max; the reference
has no beam search), ``length_penalty`` (alpha of ((5 + n) / 6) ** alpha, used to rank finished beams),
``seed`` (makes ``sample_type='sample'`` reproducible; by default a seed is drawn from torch's generator,
so ``torch.manual_seed`` governs it like it governs the reference's ``Categorical.sample``),
``noise`` (gumbel mode: ``'torch'``, the default, draws the uniform numbers with ``torch.rand`` from the global
CPU generator exactly as ``/root/reference/src/utils/gumbel.py:12`` does and copies them to the GPU, so the same
``torch.manual_seed`` gives the reference's noise; ``'device'`` draws them on the GPU).

``active_seqs`` (``:37``): a caller-provided prefix of token ids ``[B, t0]`` replaces the single BOS; results start with
it and may grow to ``t0 + max_len`` tokens (the loop still runs ``max_len`` steps, ``:155``).

``gumbel=True`` (``:83-88,104-105,136-138``): every step appends the distribution
softmax((softmax(logits) + G) / temperature) instead of a token id, and the model is fed distributions.  In the
reference this only runs when the caller passes ``active_seqs`` as distributions ``[B, 1, V]`` (ids cannot be
concatenated with them) and the concatenation is along the last axis, so a result is a ``[1, t * V]`` tensor: t
distributions back to back, the prefix first, the distribution of the step that chose EOS left out.  The same
here; without ``active_seqs`` a one-hot BOS prefix is used (the reference raises there).  Models of this
package embed a distribution as ``dist @ embedding``; any other callable gets ``[B, 1, t * V]`` as in the reference.

Any other callable works as ``model`` too (the reference's protocol ``model(inputs, active_seqs, *args,
**kwargs) -> [B, t, V]`` or ``(logits, new_inputs)`` with ``substitute_inputs``); it is then called once
per step exactly as the reference does, on GPU tensors, and only the selection / bookkeeping is ours.
"""
import torch

from .engine import DecodeLoop


class ForeignSession:
    """Steps an arbitrary model callable with the reference's protocol (``/root/reference/src/inference.py:122-132``)."""
    needs_anc = False
    takes_soft = True

    def __init__(self, model, inputs, enc_mask, kwargs, substitute_inputs, inputs_batch_dim, beam, vocab_size):
        if inputs_batch_dim not in (0, 1, 2):
            raise NotImplementedError  # as /root/reference/src/inference.py:117
        self.model, self.kwargs, self.sub, self.dim = model, kwargs, substitute_inputs, inputs_batch_dim
        self.inputs, self.enc_mask = inputs, enc_mask   # one row per sentence until the first reorder fans out
        self.V = vocab_size
        self.soft_hist = None   # gumbel mode: the distributions fed so far, [rows, t, V]
        self.pos0 = 0           # length of a caller-provided id prefix minus one (set by InferenceState)

    def step(self, rows, n, t, soft=None):
        if soft is None:
            active = rows.hist[:n, :t + 1].long()
        else:   # [n, 1, (t + 1) * V]: the distributions back to back, as /root/reference/src/inference.py:105 builds them
            new = soft[:n, :self.V].unsqueeze(1)
            self.soft_hist = new.clone() if self.soft_hist is None else torch.cat([self.soft_hist, new], dim=1)
            active = self.soft_hist.reshape(n, 1, -1)
            if self.sub:
                raise NotImplementedError('gumbel decoding of a stateful model')
        args = [] if self.enc_mask is None else self.enc_mask
        with torch.no_grad():
            if self.sub:   # /root/reference/src/inference.py:128: active_seqs[:, num_steps_done:], whatever the prefix length
                out, self.inputs = self.model(self.inputs, active[:, t - self.pos0:], *args, **self.kwargs)
            else:
                out = self.model(self.inputs, active, *args, **self.kwargs)
        logits = out[:, -1]
        if logits.shape[-1] != self.V:
            raise ValueError('model returned %d classes, vocabulary has %d' % (logits.shape[-1], self.V))
        logits = logits.to(torch.float32).contiguous()
        # The selection kernels order finite numbers (and -inf as the smallest); a NaN or +inf logit has no defined rank
        # there (torch.max would return the NaN's index).  A foreign model is the only source of such values: refuse them.
        if not bool(torch.isfinite(logits).logical_or(logits == float('-inf')).all()):
            raise FloatingPointError('the model returned NaN or +inf logits at step %d' % t)
        return logits

    def reorder(self, row_parent, n_rows, identity):
        if identity:
            return
        idx = row_parent[:n_rows].long()
        self.inputs = self.inputs.index_select(self.dim, idx)
        if self.enc_mask is not None:
            self.enc_mask = self.enc_mask.index_select(0, idx)
        if self.soft_hist is not None:
            self.soft_hist = self.soft_hist.index_select(0, idx)


class InferenceState:
    def __init__(self, state_dict: dict):
        # Required arguments (/root/reference/src/inference.py:14-16)
        self.model = state_dict['model']
        self.vocab = state_dict['vocab']
        self.inputs = state_dict['inputs']

        # Optional arguments, same defaults (/root/reference/src/inference.py:19-37)
        self.bos_token = state_dict.get('bos_token', '<bos>')
        self.eos_token = state_dict.get('eos_token', '<eos>')
        self.pad_token = state_dict.get('pad_token', '<pad>')
        self.unk_token = state_dict.get('unk_token', '<unk>')
        self.bos_idx = self.vocab.stoi[self.bos_token]
        self.eos_idx = self.vocab.stoi[self.eos_token]
        self.pad_idx = self.vocab.stoi[self.pad_token]
        self.unk_idx = self.vocab.stoi[self.unk_token]
        self.max_len = state_dict.get('max_len', 100)
        self.temperature = state_dict.get('temperature', 1)
        self.sample_type = state_dict.get('sample_type', 'max')
        self.kwargs = state_dict.get('kwargs', {})
        self.gumbel = state_dict.get('gumbel', False)
        self.enc_mask = state_dict.get('enc_mask')
        self.should_stack_finished = state_dict.get('should_stack_finished', False)
        self.inputs_batch_dim = state_dict.get('inputs_batch_dim', 0)
        self.substitute_inputs = state_dict.get('substitute_inputs', False)
        self.batch_size = self.inputs.size(self.inputs_batch_dim)
        self.active_seqs = state_dict.get('active_seqs')

        # Additive keys
        self.beam_size = state_dict.get('beam_size', 1)
        self.length_penalty = state_dict.get('length_penalty', 0.0)
        self.seed = state_dict.get('seed')  # sampling only; default: drawn from torch's global generator
        self.noise = state_dict.get('noise', 'torch')  # gumbel only: where the uniform draws come from
        # 'dtype': 'fp32' | 'bf16' selects the arithmetic of a model that can switch (TransformerLM); for any other model
        # it must name what the model computes in (fp32).  'return_logits': keep the last-position logits of every step
        # in ``self.step_logits`` as (logits [rows, V], sentence index of every row), a debugging aid.
        self.dtype = state_dict.get('dtype')
        self.return_logits = state_dict.get('return_logits', False)
        self.step_logits = []

        self.finished = [None for _ in range(self.batch_size)]
        self.num_steps_done = 0
        self.validate()

    def validate(self):
        assert self.max_len > 0
        assert self.batch_size > 0
        assert type(self.bos_token) is str
        assert type(self.eos_token) is str
        # bos/eos must be in the dictionary (/root/reference/src/inference.py:56-58)
        assert self.bos_idx != self.vocab.stoi[self.unk_token]
        assert self.eos_idx != self.vocab.stoi[self.unk_token]

    def _session(self, n_positions=None):
        if hasattr(self.model, 'session') and not self.kwargs.get('force_foreign', False):
            extra = {}
            if getattr(self.model, 'session_takes_kwargs', False):   # e.g. style= of a ConditionalLM
                extra = {k: v for k, v in self.kwargs.items() if k not in ('force_foreign', 'return_z')}
            return self.model.session(self.inputs, self.enc_mask, self.beam_size, n_positions or self.max_len, **extra)
        kwargs = {k: v for k, v in self.kwargs.items() if k != 'force_foreign'}
        return ForeignSession(self.model, self.inputs, self.enc_mask, kwargs, self.substitute_inputs,
                              self.inputs_batch_dim, self.beam_size, len(self.vocab))

    def _run(self, keep_logits=0):
        if self.return_logits:
            keep_logits = self.max_len
        restore_dtype = None
        if self.dtype is not None:
            if self.dtype not in ('fp32', 'bf16'):
                raise ValueError("dtype must be 'fp32' or 'bf16'")
            have = getattr(self.model, 'dtype', 'fp32')
            if have != self.dtype:
                if not hasattr(self.model, 'set_dtype'):
                    raise ValueError('the model computes in %s, dtype=%r was asked for' % (have, self.dtype))
                self.model.set_dtype(self.dtype)
                restore_dtype = have      # the switch lasts for this call only: the caller's model is handed back as it came
        try:   # everything after the switch: validation, session and loop set-up can raise too
            if self.sample_type not in ('max', 'sample'):
                raise NotImplementedError  # as /root/reference/src/inference.py:149
            sample = None
            if self.sample_type == 'sample':
                if self.beam_size != 1:
                    raise ValueError("sample_type='sample' needs beam_size 1")
                seed = self.seed if self.seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
                sample = (float(self.temperature), seed)
            prefix = None
            if self.active_seqs is not None and not self.gumbel:
                prefix = self.active_seqs
                if prefix.dim() != 2 or prefix.shape[0] != self.batch_size or prefix.shape[1] < 1 or prefix.is_floating_point():
                    raise ValueError('active_seqs must be token ids [batch, t0 >= 1]')
                if int(prefix.min()) < 0 or int(prefix.max()) >= len(self.vocab):   # nn.Embedding raises IndexError there
                    raise IndexError('active_seqs holds token ids outside the vocabulary (0..%d)' % (len(self.vocab) - 1))
            if not torch.cuda.is_available():
                raise RuntimeError('InferenceState needs a GPU: the decode loop runs in HIP kernels and has no CPU path')
            device = self.inputs.device if self.inputs.is_cuda else torch.device('cuda', torch.cuda.current_device())
            gumbel = None
            if self.gumbel:
                if self.beam_size != 1:
                    raise ValueError('gumbel=True needs beam_size 1')
                if self.substitute_inputs:
                    # /root/reference/src/inference.py:128 slices [B, 1, t * V] along its middle axis: empty from step 1 on
                    raise NotImplementedError('gumbel decoding of a stateful model')
                if self.noise not in ('torch', 'device'):
                    raise ValueError("noise must be 'torch' or 'device'")
                V = len(self.vocab)
                if self.active_seqs is None:
                    dist_prefix = torch.zeros((self.batch_size, V), dtype=torch.float32, device=device)
                    dist_prefix[:, self.bos_idx] = 1.0
                else:
                    if tuple(self.active_seqs.shape) != (self.batch_size, 1, V) or not self.active_seqs.is_floating_point():
                        raise NotImplementedError('gumbel=True takes active_seqs as one distribution per sentence: [B, 1, V] floats')
                    dist_prefix = self.active_seqs[:, 0].to(device=device, dtype=torch.float32)
                if self.noise == 'torch':
                    draw = lambda n, v: torch.rand((n, v)).to(device)            # noqa: E731
                else:
                    draw = lambda n, v: torch.rand((n, v), device=device)        # noqa: E731
                gumbel = (float(self.temperature), dist_prefix, draw)
            with torch.cuda.device(device):
                total = self.max_len if prefix is None else self.max_len + prefix.shape[1] - 1
                session = self._session(total)
                if prefix is not None and prefix.shape[1] > 1:
                    if isinstance(session, ForeignSession):
                        session.pos0 = prefix.shape[1] - 1
                    elif self.substitute_inputs:
                        # /root/reference/src/inference.py:128 feeds active_seqs[:, num_steps_done:], so with a prefix of t0 > 1
                        # tokens a stateful model is given t0 tokens at EVERY step (the last t0 - 1 of them again); a session
                        # that carries its state consumes each token once and would give different results
                        raise NotImplementedError('an id prefix longer than one token together with substitute_inputs is only '
                                                  'supported for foreign model callables')
                loop = DecodeLoop(session, self.batch_size, self.beam_size, self.max_len, self.bos_idx,
                                  self.eos_idx, self.pad_idx, self.length_penalty, device, sample, gumbel, prefix)
                loop.select_log = getattr(self, 'select_log', None)
                loop.run(keep_logits)
        finally:
            if restore_dtype is not None:
                self.model.set_dtype(restore_dtype)
        self.loop = loop
        self.step_logits = [(lg, sent) for lg, sent, _ in loop.kept]
        self.num_steps_done = max(0, loop.steps_done - 1)   # index of the last step, as /root/reference/src/inference.py:156 leaves it
        self.scores = loop.out_score
        return loop

    def decode_packed(self):
        """Runs the loop and returns the outputs as device tensors, without building the reference's list of
        per-sentence tensors: (tokens [B, 1 + max_len] int32, padded with ``<pad>``; lengths [B] int32, BOS
        included and EOS stripped; scores [B] fp32)."""
        loop = self._run()
        return loop.out_tok, loop.out_len, loop.out_score

    def inference(self, keep_logits=0):
        loop = self._run(keep_logits)
        lens = loop.out_len.cpu().tolist()
        if self.gumbel:
            # [1, t * V] per sentence, as the reference's last-axis concatenation leaves them
            self.finished = [loop.dist_seq[i, :n].reshape(1, -1) for i, n in enumerate(lens)]
            if self.should_stack_finished:
                # every sequence has length 1 along dim 0, so /root/reference/src/inference.py:71-78 pads nothing
                # and stacks; torch.stack raises for sequences of different numbers of steps, as it does there
                self.finished = torch.stack(self.finished)
            return self.finished
        toks = loop.out_tok.long()
        self.finished = [toks[i, :n] for i, n in enumerate(lens)]
        if self.should_stack_finished:
            self.stack_finished()
        return self.finished

    def stack_finished(self):
        """Pads with ``<pad>`` and stacks (``/root/reference/src/inference.py:71-92``).  The reference builds its
        pads as float zeros, so the stacked tensor is float32 under type promotion; kept for parity."""
        width = max(len(s) for s in self.finished)
        rows = []
        for s in self.finished:
            pads = torch.zeros(width - len(s), device=s.device).fill_(self.pad_idx)
            rows.append(s if len(s) == width else torch.cat((s, pads), dim=0))
        self.finished = torch.stack(rows)


def simple_inference(model, z, vocab, max_len=100, eos_token='<eos>'):
    """``/root/reference/src/inference.py:170-181``: greedy decode, result as ``list[list[int]]``."""
    state = InferenceState({'model': model, 'inputs': z, 'vocab': vocab, 'max_len': max_len, 'eos_token': eos_token})
    loop = state._run(0)                     # the packed result: two device-to-host copies for the whole batch
    lens = loop.out_len.cpu().tolist()
    toks = loop.out_tok.cpu().tolist()
    return [row[:n] for row, n in zip(toks, lens)]
