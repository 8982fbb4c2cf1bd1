"""CPU: ``itos_many`` and corpus BLEU against outputs of the reference's own functions
(``/root/reference/src/utils/data_utils.py:38-50``, ``/root/reference/src/losses/bleu.py:45-115``) recorded in
tests/golden/postprocess.json by tests/golden/make_golden.py."""
import json
import os

import pytest
import torch

from helpers import GOLDEN


@pytest.fixture(scope='module')
def fx():
    return json.load(open(os.path.join(GOLDEN, 'postprocess.json')))


def test_itos_many_matches_reference(fx):
    import tst_amd
    vocab = tst_amd.Vocab(fx['vocab'][4:])
    assert vocab.itos == fx['vocab']
    assert tst_amd.itos_many(fx['seqs'], vocab) == fx['itos_default']
    assert tst_amd.itos_many(fx['seqs'], vocab, remove_special=False) == fx['itos_keep_special']
    assert tst_amd.itos_many(fx['seqs'], vocab, sep='') == fx['itos_chars']
    as_tensors = [torch.tensor(s, dtype=torch.long) for s in fx['seqs']]       # what InferenceState returns
    assert tst_amd.itos_many(as_tensors, vocab) == fx['itos_default']


def test_bleu_matches_reference(fx):
    import tst_amd
    b = fx['bleu']
    assert tst_amd.compute_bleu_for_sents(b['hyps'], b['refs']) == pytest.approx(b['plain'], rel=1e-12)
    assert tst_amd.compute_bleu_for_sents(b['hyps'], b['refs'], smooth=True) == pytest.approx(b['smooth'], rel=1e-12)
    assert tst_amd.compute_bleu_for_sents(b['hyps'], b['refs'], max_order=2) == pytest.approx(b['order2'], rel=1e-12)
    full = tst_amd.corpus_bleu([[r.split()] for r in b['refs']], [h.split() for h in b['hyps']])
    assert full[0] == pytest.approx(b['full'][0], rel=1e-12) and list(full[1]) == pytest.approx(b['full'][1])
    assert list(full[2:]) == pytest.approx(b['full'][2:])
    per = [tst_amd.compute_bleu_for_sents([h], [r], smooth=True) for h, r in zip(b['hyps'], b['refs'])]
    assert per == pytest.approx(b['per_sentence'], rel=1e-12)
    multi = tst_amd.corpus_bleu([[b['refs'][0].split(), b['refs'][1].split()], [b['refs'][2].split()]],
                                [b['hyps'][0].split(), b['hyps'][2].split()])
    assert multi[0] == pytest.approx(b['multi_ref'][0], rel=1e-12) and list(multi[1]) == pytest.approx(b['multi_ref'][1])


def test_bleu_edge_cases():
    import tst_amd
    assert tst_amd.compute_bleu_for_sents(['a b c d e'], ['a b c d e']) == pytest.approx(1.0)
    assert tst_amd.compute_bleu_for_sents(['x y'], ['a b c d']) == 0
    assert tst_amd.compute_bleu_for_sents([''], ['a b']) == 0
