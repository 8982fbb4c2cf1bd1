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
    assert tst_amd.version() == 107


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
