"""CPU: the C-ABI library loads and exports every symbol that include/tst_hip.h declares; the host-side
argument checks reject bad calls without touching a GPU."""
import ctypes
import os
import re

import pytest

from helpers import ROOT

LIB = os.path.join(ROOT, 'text-style-transfer_amd', 'libtst_hip.so')


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'tst_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tst_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), 'build it first: text-style-transfer_amd/csrc/build.sh'
    dll = ctypes.CDLL(LIB)
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(dll, n), n


def test_binding_covers_header_and_version():
    import tst_amd
    from tst_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_symbols()
    assert tst_amd.version() == 109


def test_argument_errors_are_reported_not_crashed():
    from tst_amd import _lib
    with pytest.raises(_lib.TstError, match='null pointer'):
        _lib.call('tst_gemm_mfma_f32', None, 4, None, 4, None, None, 0, None, 4, 1, 4, 4, 0, None)
    with pytest.raises(_lib.TstError, match='multiples of 4'):
        _lib.call('tst_gemm_mfma_f32', 16, 6, 16, 4, None, None, 0, 16, 4, 1, 4, 6, 0, None)
    with pytest.raises(_lib.TstError, match='1 <= k'):
        _lib.call('tst_logsoftmax_topk_f32', 16, 100, 1, 100, 9, 16, 16, 16, None)
    with pytest.raises(_lib.TstError, match='tst_ln_gemm_f32: null pointer'):
        _lib.call('tst_ln_gemm_f32', 16, 16, 16, 1e-5, None, 16, 4, None, 16, 4, 1, 4, 4, 0, None)
    with pytest.raises(_lib.TstError, match='strip must be'):
        _lib.call('tst_gemm_skinny_f32', 16, 4, 16, 4, None, None, 0, 16, 4, 1, 4, 4, 0, 3, None)
    # the whole-step launcher checks its plan before the first launch
    with pytest.raises(_lib.TstError, match='null plan'):
        _lib.call('tst_tfm_step_run', None, None, None, None, 1, 0, 7, None)
    layers = (_lib.TstTfmLayer * 1)()
    plan = _lib.TstTfmStep(n_layers=1, d=8, heads=2, d_ff=16, V=10, n_pos=4, slab_rows=2, layers=layers)
    with pytest.raises(_lib.TstError, match='buffers hold 2'):
        _lib.call('tst_tfm_step_run', plan, None, None, None, 3, 0, 7, None)
    with pytest.raises(_lib.TstError, match='position 4 outside'):
        _lib.call('tst_tfm_step_run', plan, None, None, None, 2, 4, 7, None)
    with pytest.raises(_lib.TstError, match='bad phases'):
        _lib.call('tst_tfm_step_run', plan, None, None, None, 2, 0, 8, None)
    with pytest.raises(_lib.TstError, match='null plan'):
        _lib.call('tst_gru_step_run', None, None, None, 1, 7, None)
    gl = (_lib.TstGruLayer * 1)()
    gplan = _lib.TstGruStep(n_layers=1, E=8, H=8, V=10, rows=2, layers=gl)
    with pytest.raises(_lib.TstError, match='buffers hold 2'):
        _lib.call('tst_gru_step_run', gplan, None, None, 3, 7, None)
    with pytest.raises(_lib.TstError, match='null pointer'):
        _lib.call('tst_gru_reorder_run', gplan, None, 1, None)
    rows = _lib.TstRows()
    with pytest.raises(_lib.TstError, match='bad sizes'):
        _lib.call('tst_select_step_run', _lib.TstSelectStep(k=9), rows, rows, 1, 1, 0, 0, 0, None)
    with pytest.raises(_lib.TstError, match='bad sizes'):
        _lib.call('tst_select_step_run', _lib.TstSelectStep(k=4), rows, rows, 1, 2, 0, 0, 0, None)


def test_argmax_keys_entry_points_and_key_decoding():
    """The fused LayerNorm + projection + arg-max entry point refuses bad calls on the host, its shape predicate is a pure
    function, a step plan with keys refuses calls the fused kernel cannot serve, and the host-side decoding of keys follows
    the documented format (unsigned maximum over a row's slots; 0 = no candidate = token 0; else ~low word)."""
    import torch
    from tst_amd import _lib, kernels as K
    assert _lib._dll.tst_vocab_argmax_ok(256, 32000, 256) == 1
    assert _lib._dll.tst_vocab_argmax_ok(257, 32000, 256) == 0      # 3 x 250 tiles of 128 x 128
    assert _lib._dll.tst_vocab_argmax_ok(8, 100, 384) == 0          # rows of 256 or 512 floats only
    assert _lib._dll.tst_vocab_argmax_ok(0, 100, 256) == 0
    with pytest.raises(_lib.TstError, match='tst_ln_vocab_argmax_f32: null pointer'):
        _lib.call('tst_ln_vocab_argmax_f32', 16, 16, 16, 1e-5, 16, 100, None, None, 4, 100, 256, None)
    with pytest.raises(_lib.TstError, match='ldw'):
        _lib.call('tst_ln_vocab_argmax_f32', 16, 16, 16, 1e-5, 16, 98, None, 16, 4, 100, 256, None)
    with pytest.raises(_lib.TstError, match='tst_vocab_argmax_ok'):
        _lib.call('tst_ln_vocab_argmax_f32', 16, 16, 16, 1e-5, 16, 100, None, 16, 4, 100, 128, None)
    layers = (_lib.TstTfmLayer * 1)()
    plan = _lib.TstTfmStep(n_layers=1, d=256, heads=2, d_ff=16, V=10, n_pos=4, slab_rows=2, layers=layers, x=16, h=16, a=16, ff=16,
                           logits=16, argmax_keys=16)
    with pytest.raises(_lib.TstError, match='argmax_keys needs'):
        _lib.call('tst_tfm_step_run', plan, None, None, None, 2, 0, 6, None)       # not all three phases
    plan.bf16 = 1
    with pytest.raises(_lib.TstError, match='argmax_keys needs'):
        _lib.call('tst_tfm_step_run', plan, None, None, None, 2, 0, 7, None)       # bf16 plan
    rows = _lib.TstRows()
    with pytest.raises(_lib.TstError, match='am_keys and sampling'):
        _lib.call('tst_select_step_run', _lib.TstSelectStep(k=1, sample=1, am_keys=16), rows, rows, 1, 1, 0, 0, 0, None)

    def key(bits, col):
        u = (~bits & 0xffffffff) if bits & 0x80000000 else bits | 0x80000000
        v = (u << 32) | (~col & 0xffffffff)
        return v - (1 << 64) if v >= (1 << 63) else v          # as stored in an int64 tensor
    keys = torch.zeros((4, K.ARGMAX_SLOTS), dtype=torch.int64)
    keys[0, 3], keys[0, 9] = key(0x3f800000, 17), key(0x40000000, 5)     # 1.0 at 17, 2.0 at 5 -> 5
    keys[1, 0], keys[1, 15] = key(0xbf800000, 7), key(0xc0000000, 2)     # -1.0 at 7, -2.0 at 2 -> 7
    keys[2, 4], keys[2, 5] = key(0x3f800000, 900), key(0x3f800000, 31)   # equal values -> the smaller column
    assert K.tokens_of_keys(keys).tolist() == [5, 7, 31, 0]              # row 3: no candidate -> token 0
    keys[3, 1] = key(0xbf800000, 12)                                       # a negative value beats "no candidate"
    keys[3, 2] = key(0x00000000, 40)                                       # and +0 beats it
    assert K.tokens_of_keys(keys).tolist() == [5, 7, 31, 40]


def test_gi_table_entry_points_check_their_arguments():
    from tst_amd import _lib
    with pytest.raises(_lib.TstError, match='tst_gru_cell_fused_tab_f32: null pointer'):
        _lib.call('tst_gru_cell_fused_tab_f32', None, 192, 16, 10, 16, 64, 16, 192, None, 32, 64, None, 0, 4, 64, None)
    with pytest.raises(_lib.TstError, match='H %% 64'.replace('%%', '%')):
        _lib.call('tst_gru_cell_fused_tab_f32', 16, 96, 16, 10, 16, 32, 16, 96, None, 32, 32, None, 0, 4, 32, None)
    with pytest.raises(_lib.TstError, match='leading dimensions'):
        _lib.call('tst_gru_cell_fused_tab_f32', 16, 100, 16, 10, 16, 64, 16, 192, None, 32, 64, None, 0, 4, 64, None)
    with pytest.raises(_lib.TstError, match='must not overlap'):
        _lib.call('tst_gru_cell_fused_tab_f32', 16, 192, 16, 10, 1024, 64, 16, 192, None, 1024, 64, None, 0, 4, 64, None)
    with pytest.raises(_lib.TstError, match='tst_gru_cell_step_tab_f32: null pointer'):
        _lib.call('tst_gru_cell_step_tab_f32', 16, 192, None, 10, 16, 16, 16, 4, 64, None)
    with pytest.raises(_lib.TstError, match='ld_tab'):
        _lib.call('tst_gru_cell_step_tab_f32', 16, 100, 16, 10, 16, 16, 16, 4, 64, None)
    gl = (_lib.TstGruLayer * 1)()
    gplan = _lib.TstGruStep(n_layers=1, E=8, H=8, V=10, rows=2, layers=gl, x=16, gi=16, gh=16, logits=16, gi_tab=16, ld_gi_tab=8)
    with pytest.raises(_lib.TstError, match='gi_tab needs tok'):
        _lib.call('tst_gru_step_run', gplan, None, None, 2, 7, None)


def test_sizes_beyond_31_bits_are_refused_by_every_entry_point():
    """Row counts, widths and vocabulary sizes above 2^31 - 1 are refused on the host (the kernels count them in 32-bit integers);
    leading dimensions may be larger.  Random argument lists with one such size in them never reach a launch."""
    import ctypes
    import random
    import re
    from tst_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'tst_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    rnd = random.Random(3)
    checked = 0
    for name, (args, res) in _lib._PROTOS.items():
        m = re.search(r'\bint\s+%s\s*\(([^)]*)\)' % name, header)
        if res is not ctypes.c_int or name.endswith('_ok') or m is None:
            continue
        params = [q.strip() for q in m.group(1).split(',')]
        if len(params) != len(args) or any('tst_' in q for q in params):        # plans are checked through their own tests
            continue
        sizes = [i for i, q in enumerate(params) if q.startswith('int64_t')
                 and not re.match(r'(ld|.*_ld$|.*stride|pad$)', q.split()[-1])]
        for i in sizes:
            call = []
            for j, (t, q) in enumerate(zip(args, params)):
                if t is _lib.c_vp:
                    call.append(4096)
                elif t is _lib.c_f32:
                    call.append(1.0)
                else:
                    call.append(rnd.choice([4, 8, 64, 256]))
            call[i] = rnd.choice([1 << 31, 1 << 40])
            call[-1] = None
            rc = getattr(_lib._dll, name)(*call)
            assert rc in (-1, -3), (name, params[i], rc)
            checked += 1
    assert checked >= 90
    with pytest.raises(_lib.TstError, match='exceeds 2\\^31 - 1'):
        _lib.call('tst_layernorm_f32', 16, 16, 16, 1e-5, 16, 4, 1 << 31, None)


def test_no_product_module_touches_the_oracle():
    pkg = os.path.join(ROOT, 'text-style-transfer_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.sh')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in text, os.path.join(dirpath, f)


def test_inference_state_validation_matches_reference_asserts():
    """Same constructor checks as /root/reference/src/inference.py:50-58 (no GPU needed to construct)."""
    import torch
    import tst_amd
    vocab = tst_amd.Vocab.synthetic(16)
    z = torch.zeros(3, 8)
    with pytest.raises(AssertionError):
        tst_amd.InferenceState({'model': None, 'inputs': z, 'vocab': vocab, 'max_len': 0})
    with pytest.raises(AssertionError):
        tst_amd.InferenceState({'model': None, 'inputs': z, 'vocab': vocab, 'eos_token': 'not-in-vocab'})
    with pytest.raises(KeyError):
        tst_amd.InferenceState({'model': None, 'vocab': vocab})
    st = tst_amd.InferenceState({'model': None, 'inputs': z, 'vocab': vocab})
    assert (st.batch_size, st.max_len, st.bos_idx, st.eos_idx, st.sample_type) == (3, 100, 2, 3, 'max')
