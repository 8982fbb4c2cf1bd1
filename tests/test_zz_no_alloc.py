"""GPU: ``tst_retire_compact_noalloc`` (alive flags and prefix sums in an array of the library's code object, no workspace
argument, no allocation) against ``tst_retire_compact`` (caller's workspace) through the C ABI: identical outputs, bit for
bit, on two calls in a row, the second with a smaller input (the array keeps the first call's contents)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _random_state(n_sent, k, T, step, seed, eos=3):
    """A consistent random selection state after `step` steps: (cur rows, sel arrays, lp) on the GPU."""
    from tst_amd import kernels as K
    g = torch.Generator().manual_seed(seed)
    R = n_sent * k
    dev = torch.device('cuda', 0)
    cur = K.RowState(R, T, dev, True, 1)
    fin = (torch.rand(R, generator=g) < 0.5).to(torch.int32)
    nemit = torch.where(fin != 0, torch.randint(1, step + 1, (R,), generator=g, dtype=torch.int32), torch.full((R,), step, dtype=torch.int32))
    cur.hist.copy_(torch.randint(4, 1000, (R, T + 1), generator=g, dtype=torch.int32))
    cur.anc.copy_(torch.randint(0, R, (R, T), generator=g, dtype=torch.int32))
    cur.sent.copy_(torch.arange(n_sent, dtype=torch.int32).repeat_interleave(k))
    cur.tok.copy_(torch.randint(4, 1000, (R,), generator=g, dtype=torch.int32))
    cur.score.copy_(-torch.rand(R, generator=g) * 10)
    cur.fin.copy_(fin)
    cur.nemit.copy_(nemit)
    sel = K.Selection(R, dev)
    parent = (torch.arange(n_sent).repeat_interleave(k) * k + torch.randint(0, k, (R,), generator=g)).to(torch.int32)
    pfin, pnemit = fin[parent.long()], nemit[parent.long()]
    tok = torch.randint(3, 40, (R,), generator=g, dtype=torch.int32)     # eos (3) now and then
    sfin = ((pfin != 0) | (tok == eos)).to(torch.int32)
    sel.parent.copy_(parent)
    sel.tok.copy_(tok)
    sel.fin.copy_(sfin)
    sel.nemit.copy_(pnemit + (pfin == 0).to(torch.int32))
    sel.score.copy_(-torch.rand(R, generator=g) * 10)
    lp = torch.tensor([((5.0 + n) / 6.0) ** 0.6 for n in range(T + 2)], dtype=torch.float32, device=dev)
    return cur, sel, lp


def _run(entry, cur, sel, lp, n_sent, k, T, step, with_scratch):
    from tst_amd import _lib, kernels as K
    dev = torch.device('cuda', 0)
    R = n_sent * k
    nxt = K.RowState(R, T, dev, True, 1)
    i32 = dict(dtype=torch.int32, device=dev)
    out_tok = torch.full((n_sent, T + 1), 1, **i32)
    out_len = torch.zeros(n_sent, **i32)
    out_score = torch.zeros(n_sent, dtype=torch.float32, device=dev)
    row_parent = torch.zeros(R, **i32)
    n_alive = torch.zeros(1, **i32)
    scratch = torch.zeros(2 * n_sent, **i32)
    args = [cur.struct(), nxt.struct(), *sel.ptrs(), lp.data_ptr(), n_sent, k, step, T, 3, 0, 0, out_tok.data_ptr(), out_len.data_ptr(),
            out_score.data_ptr()]
    if with_scratch:
        args.append(scratch.data_ptr())
    args += [row_parent.data_ptr(), n_alive.data_ptr(), K._stream()]
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(dev)[0]
    _lib.call(entry, *args)
    torch.cuda.synchronize()
    free_delta = torch.cuda.mem_get_info(dev)[0] - free_before      # device memory taken (or given back) by the call itself
    return dict(free_delta=free_delta, hist=nxt.hist, anc=nxt.anc, sent=nxt.sent, tok=nxt.tok, score=nxt.score, fin=nxt.fin, nemit=nxt.nemit, out_tok=out_tok,
                out_len=out_len, out_score=out_score, row_parent=row_parent, n_alive=n_alive)


def test_noalloc_entry_equals_workspace_entry_on_two_calls_the_second_smaller():
    torch.cuda.set_device(0)
    k, T, step = 4, 20, 7
    for call, (n_sent, seed) in enumerate(((3000, 1), (1100, 2))):   # both beyond the one-workgroup route, so the array is used
        cur, sel, lp = _random_state(n_sent, k, T, step, seed)
        want = _run('tst_retire_compact', cur, sel, lp, n_sent, k, T, step, True)
        got = _run('tst_retire_compact_noalloc', cur, sel, lp, n_sent, k, T, step, False)
        alive = int(want['n_alive'])
        assert 0 < alive < n_sent, 'the case must retire some sentences and keep some'
        for name in want:
            if name != 'free_delta':
                assert torch.equal(got[name], want[name]), (n_sent, name)
        if call == 1:   # the first call on a device resolves the array's address (it may load the code object); later calls
            assert got['free_delta'] == 0, 'tst_retire_compact_noalloc changed the free device memory by %d bytes' % got['free_delta']


def test_noalloc_entry_rejects_sizes_beyond_its_array():
    from tst_amd import _lib
    rows = _lib.TstRows()
    with pytest.raises(_lib.TstError, match='the built-in scratch holds 65536'):
        _lib.call('tst_retire_compact_noalloc', rows, rows, None, None, None, None, None, None, 65537, 1, 0, 4, 3, 0, 0, None, None, None,
                  None, None, None)
