"""ctypes binding of ``libtst_hip.so`` (declared in ``include/tst_hip.h``).

The library is built in-tree by ``csrc/build.sh`` (``__graft_entry__.build()``).  There is no fallback:
if it is missing, importing this module raises, and every wrapper raises on a non-zero status.
"""
import ctypes
import os

# torch must be loaded first: the library's dependency on libamdhip64.so.7 then binds to the HIP runtime
# that torch already loaded, so both share one runtime (device pointers, streams).  Loaded the other way
# round the process ends up with two HIP runtimes and the second one finds no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# TST_HIP_LIB names another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get('TST_HIP_LIB') or os.path.join(_HERE, 'libtst_hip.so')

c_i64, c_i32, c_f32, c_vp, c_int = ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_int


class TstRows(ctypes.Structure):
    """``tst_rows`` of include/tst_hip.h"""
    _fields_ = [(n, c_vp) for n in ('hist', 'anc', 'sent', 'tok', 'score', 'fin', 'nemit')]


class TstTfmLayer(ctypes.Structure):
    """``tst_tfm_layer`` of include/tst_hip.h"""
    _fields_ = [('ln_self_g', c_vp), ('ln_self_b', c_vp),
                ('wqkv', c_vp), ('bqkv', c_vp), ('ld_wqkv', c_i64),
                ('wo', c_vp), ('bo', c_vp), ('ld_wo', c_i64),
                ('ln_cross_g', c_vp), ('ln_cross_b', c_vp),
                ('cwq', c_vp), ('cbq', c_vp), ('ld_cwq', c_i64),
                ('cwo', c_vp), ('cbo', c_vp), ('ld_cwo', c_i64),
                ('ln_ff_g', c_vp), ('ln_ff_b', c_vp),
                ('w1', c_vp), ('b1', c_vp), ('ld_w1', c_i64),
                ('w2', c_vp), ('b2', c_vp), ('ld_w2', c_i64),
                ('cache', c_vp), ('kv_cross', c_vp)]


class TstTfmStep(ctypes.Structure):
    """``tst_tfm_step`` of include/tst_hip.h"""
    _fields_ = [('n_layers', c_i32), ('d', c_i32), ('heads', c_i32), ('d_ff', c_i32), ('V', c_i32), ('bf16', c_i32),
                ('group', c_i32), ('reserved', c_i32), ('n_pos', c_i64), ('slab_rows', c_i64), ('S', c_i64), ('anc_ld', c_i64),
                ('emb', c_vp), ('pe', c_vp), ('scale', c_f32), ('eps', c_f32),
                ('z', c_vp), ('mask', c_vp), ('kvrow', c_vp), ('ln_g', c_vp), ('ln_b', c_vp),
                ('wout', c_vp), ('bout', c_vp), ('ld_wout', c_i64),
                ('x', c_vp), ('h', c_vp), ('q', c_vp), ('a', c_vp), ('ff', c_vp), ('logits', c_vp), ('ld_logits', c_i64),
                ('layers', ctypes.POINTER(TstTfmLayer)), ('ht', c_vp), ('ld_ht', c_i64), ('part', c_vp), ('n_slices', c_i64),
                ('argmax_keys', c_vp)]


class TstGruLayer(ctypes.Structure):
    """``tst_gru_layer`` of include/tst_hip.h"""
    _fields_ = [('wih', c_vp), ('bih', c_vp), ('ld_wih', c_i64), ('whh', c_vp), ('bhh', c_vp), ('ld_whh', c_i64),
                ('h', c_vp), ('h_alt', c_vp)]


class TstGruStep(ctypes.Structure):
    """``tst_gru_step`` of include/tst_hip.h"""
    _fields_ = [('n_layers', c_i32), ('E', c_i32), ('H', c_i32), ('V', c_i32), ('rows', c_i64),
                ('emb', c_vp), ('extra', c_vp), ('wout', c_vp), ('bout', c_vp), ('ld_wout', c_i64),
                ('x', c_vp), ('gi', c_vp), ('gh', c_vp), ('logits', c_vp), ('ld_logits', c_i64),
                ('layers', ctypes.POINTER(TstGruLayer)), ('ht', c_vp), ('ld_ht', c_i64), ('gi_tab', c_vp), ('ld_gi_tab', c_i64)]


class TstSelectStep(ctypes.Structure):
    """``tst_select_step`` of include/tst_hip.h"""
    _fields_ = [('k', c_i32), ('eos', c_i32), ('V', c_i32), ('bf16_logits', c_i32), ('logits', c_vp), ('ld_logits', c_i64),
                ('am', c_vp), ('top_val', c_vp), ('top_idx', c_vp), ('lse', c_vp),
                ('sel_parent', c_vp), ('sel_tok', c_vp), ('sel_score', c_vp), ('sel_fin', c_vp), ('sel_nemit', c_vp),
                ('lp', c_vp), ('T', c_i64), ('top_ends', c_i32), ('sample', c_i32), ('temperature', c_f32),
                ('seed', ctypes.c_uint32),
                ('out_tok', c_vp), ('out_len', c_vp), ('out_score', c_vp), ('scratch_alive_pos', c_vp), ('row_parent', c_vp),
                ('n_alive', c_vp), ('n_sent_dev', c_vp), ('part', c_vp), ('n_slices', c_i64), ('stats_k', c_i64),
                ('am_keys', c_vp), ('n_alive_host', c_vp)]


STEP_EMBED, STEP_LAYERS, STEP_VOCAB, STEP_FUSED_CELL = 1, 2, 4, 8

_PROTOS = {
    'tst_version': ([], c_int),
    'tst_last_error': ([], ctypes.c_char_p),
    'tst_embed_posadd_f32': ([c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_embed_posadd_rows_f32': ([c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_scale_posadd_f32': ([c_vp, c_f32, c_vp, c_i64, c_i64, c_vp], c_int),
    'tst_layernorm_f32': ([c_vp, c_vp, c_vp, c_f32, c_vp, c_i64, c_i64, c_vp], c_int),
    'tst_gemm_mfma_f32': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp], c_int),
    'tst_gemm_dma_f32': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp], c_int),
    'tst_layernorm_t_f32': ([c_vp, c_vp, c_vp, c_f32, c_vp, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_gemm_skinny_f32': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_vp], c_int),
    'tst_ln_gemm_f32': ([c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp], c_int),
    'tst_vocab_argmax_ok': ([c_i64, c_i64, c_i64], c_int),
    'tst_ln_vocab_argmax_f32': ([c_vp, c_vp, c_vp, c_f32, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_chain_ok': ([c_i64, c_i64, c_i64, c_i64], c_int),
    'tst_chain_f32': ([c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_i64, c_vp, c_int, c_vp, c_i64, c_i64, c_vp, c_i64,
                       c_i64, c_i64, c_vp], c_int),
    'tst_attn_self_step_f32': ([c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_attn_cross_step_f32': ([c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_attn_cross_packed_f32': ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_attn_cross_group_f32': ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_attn_cross_mfma_f32': ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_gru_cell_step_f32': ([c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp], c_int),
    'tst_gru_cell_fused_ok': ([c_i64, c_i64], c_int),
    'tst_gru_cell_fused_tab_f32': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_gru_cell_step_tab_f32': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp], c_int),
    'tst_gru_cell_fused_f32': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_gru_cell_fused_t_f32': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64,
                                  c_vp], c_int),
    'tst_gather_rows_f32': ([c_vp, c_vp, c_vp, c_i64, c_i64, c_vp], c_int),
    'tst_scatter_rows_f32': ([c_vp, c_vp, c_vp, c_i64, c_i64, c_vp], c_int),
    'tst_pack_sources': ([c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp], c_int),
    'tst_argmax_rows_f32': ([c_vp, c_i64, c_i64, c_i64, c_vp, c_vp], c_int),
    'tst_gumbel_softmax_f32': ([c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_f32, c_vp, c_i64, c_vp], c_int),
    'tst_mix_logprobs_f32': ([c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp], c_int),
    'tst_sample_rows_f32': ([c_vp, c_i64, c_i64, c_i64, c_f32, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp, c_vp], c_int),
    'tst_logsoftmax_topk_f32': ([c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp], c_int),
    'tst_layernorm_f32_bf16': ([c_vp, c_vp, c_vp, c_f32, c_vp, c_i64, c_i64, c_vp], c_int),
    'tst_add_rows_f32': ([c_vp, c_vp, c_vp, c_i64, c_i64, c_vp], c_int),
    'tst_gemm_mfma_bf16': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_vp],
                           c_int),
    'tst_gemm_res_ln_ok': ([c_i64, c_i64, c_i64], c_int),
    'tst_gemm_res_ln_bf16': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_vocab_stats_ok': ([c_i64, c_i64, c_i64], c_int),
    'tst_vocabproj_stats_bf16': ([c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp], c_int),
    'tst_topk_from_stats_bf16': ([c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp], c_int),
    'tst_attn_self_step_bf16': ([c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_attn_cross_step_bf16': ([c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp], c_int),
    'tst_argmax_rows_bf16': ([c_vp, c_i64, c_i64, c_i64, c_vp, c_vp], c_int),
    'tst_sample_rows_bf16': ([c_vp, c_i64, c_i64, c_i64, c_f32, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp, c_vp], c_int),
    'tst_logsoftmax_topk_bf16': ([c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp], c_int),
    'tst_beam_merge': ([ctypes.POINTER(TstRows), c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                        c_vp], c_int),
    'tst_greedy_select': ([c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(TstRows), c_vp], c_int),
    'tst_tfm_step_run': ([ctypes.POINTER(TstTfmStep), c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_vp], c_int),
    'tst_gru_step_run': ([ctypes.POINTER(TstGruStep), c_vp, c_vp, c_i64, c_int, c_vp], c_int),
    'tst_gru_reorder_run': ([ctypes.POINTER(TstGruStep), c_vp, c_i64, c_vp], c_int),
    'tst_select_step_run': ([ctypes.POINTER(TstSelectStep), ctypes.POINTER(TstRows), ctypes.POINTER(TstRows), c_i64, c_i64, c_i64,
                             ctypes.c_uint32, c_int, c_vp], c_int),
    'tst_greedy_retire_compact': ([c_vp, ctypes.POINTER(TstRows), ctypes.POINTER(TstRows), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                   c_i64, c_i64, c_i32, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp], c_int),
    'tst_retire_compact': ([ctypes.POINTER(TstRows), ctypes.POINTER(TstRows), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                            c_i64, c_i64, c_i64, c_i32, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp], c_int),
    'tst_retire_compact_noalloc': ([ctypes.POINTER(TstRows), ctypes.POINTER(TstRows), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                    c_i64, c_i64, c_i64, c_i32, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp], c_int),
}

EXPORTS = tuple(_PROTOS)

if not os.path.exists(LIB_PATH):
    raise ImportError('%s is missing: build it with text-style-transfer_amd/csrc/build.sh (hipcc, gfx950); '
                      'there is no non-HIP fallback' % LIB_PATH)
_dll = ctypes.CDLL(LIB_PATH)
for _name, (_args, _res) in _PROTOS.items():
    _fn = getattr(_dll, _name)
    _fn.argtypes, _fn.restype = _args, _res


class TstError(RuntimeError):
    pass


def call(name, *args):
    """Calls an entry point and raises ``TstError`` with the library's message on a non-zero status."""
    rc = getattr(_dll, name)(*args)
    if rc != 0:
        raise TstError('%s failed (%d): %s' % (name, rc, _dll.tst_last_error().decode()))


def version():
    return _dll.tst_version()
