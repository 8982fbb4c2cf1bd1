"""The validation chain of ``/root/reference/src/trainers/transformer_mt_trainer.py:91-101``: ``transformer.encoder(src)`` ->
``InferenceState(...).inference()`` -> ``itos_many(preds)`` / ``itos_many(trg)`` -> ``compute_bleu_for_sents``.

Fixture ``tests/golden/chain_tfm.json`` (``tests/golden/make_golden.py::chain_case``) holds what the REAL loop, the REAL
``itos_many`` and the REAL ``compute_bleu_for_sents`` of the reference produced for a specification transformer over a
vocabulary with BPE continuation markers.  The GPU test runs the same chain on the loop's device outputs; the CPU test runs
the string half on the recorded token ids."""
import json
import os

import pytest
import torch

from helpers import GOLDEN, torch_model_for


@pytest.fixture(scope='module')
def fx():
    return json.load(open(os.path.join(GOLDEN, 'chain_tfm.json')))


def test_string_half_of_the_chain_on_recorded_ids(fx):
    import tst_amd
    vocab = tst_amd.Vocab(fx['words'])
    assert tst_amd.itos_many(fx['preds'], vocab) == fx['pred_str']
    assert tst_amd.itos_many(torch.tensor(fx['trg']), vocab) == fx['trg_str']
    assert tst_amd.compute_bleu_for_sents(fx['pred_str'], fx['trg_str']) == pytest.approx(fx['bleu'], rel=1e-12)
    assert 0.05 < fx['bleu'] < 0.95 and any('@@' in w for w in fx['words']) and '' in fx['pred_str']


@pytest.mark.gpu
def test_decode_itos_bleu_chain_on_device_outputs(fx):
    import tst_amd
    cfg = fx['config']
    tmodel, _ = torch_model_for(cfg)
    vocab = tst_amd.Vocab(fx['words'])
    assert len(vocab) == cfg['V']
    model = tst_amd.Transformer(tst_amd.HP(cfg['hp']), vocab, vocab)
    model.load_state_dict(tmodel.state_dict())
    src = torch.tensor(fx['src'], dtype=torch.long).cuda()
    trg = torch.tensor(fx['trg'], dtype=torch.long).cuda()
    encs, enc_mask = model.encoder(src)                                         # transformer_mt_trainer.py:91
    preds = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab,
                                    'max_len': cfg['max_len']}).inference()     # :92-98
    assert all(p.is_cuda and p.dtype == torch.long for p in preds)
    assert [p.tolist() for p in preds] == fx['preds']
    pred_str = tst_amd.itos_many(preds, vocab)                                  # :99, on the loop's device tensors
    trg_str = tst_amd.itos_many(trg, vocab)                                     # :100
    assert pred_str == fx['pred_str'] and trg_str == fx['trg_str']
    bleu = tst_amd.compute_bleu_for_sents(pred_str, trg_str)                    # :101
    assert bleu == pytest.approx(fx['bleu'], rel=1e-12)
    assert tst_amd.compute_bleu_for_sents(pred_str, trg_str, smooth=True) == pytest.approx(fx['bleu_smooth'], rel=1e-12)
    # the packed outputs (one device-to-host copy for the batch) feed the same chain
    st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab, 'max_len': cfg['max_len']})
    tok, lens, _ = st.decode_packed()
    rows = [r[:n] for r, n in zip(tok.cpu().tolist(), lens.cpu().tolist())]
    assert tst_amd.itos_many(rows, vocab) == fx['pred_str']
