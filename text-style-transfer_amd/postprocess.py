"""The step right after the decode loop: ids -> strings and corpus BLEU over the decoded sentences.

``itos_many`` follows ``/root/reference/src/utils/data_utils.py:34-50`` (drop ``<bos>/<eos>/<pad>``, join with
``sep``, undo BPE by deleting ``'@@ '``); ``compute_bleu_for_sents`` follows the corpus BLEU used at
``/root/reference/src/trainers/transformer_mt_trainer.py:101`` (``/root/reference/src/losses/bleu.py:45-115``:
clipped n-gram precisions up to order 4, geometric mean, brevity penalty exp(1 - 1/ratio), optional add-one
smoothing).  Host-side string work: it consumes the one device-to-host copy of the finished sequences.
"""
import math

SPECIAL_TOKENS = frozenset(('<bos>', '<eos>', '<pad>'))  # /root/reference/src/utils/constants.py:5


def _as_lists(seqs):
    if hasattr(seqs, 'tolist') and not isinstance(seqs, list):
        return seqs.tolist()
    return [s.tolist() if hasattr(s, 'tolist') else list(s) for s in seqs]


def itos_many(seqs, vocab, remove_special=True, sep=' '):
    """Sequences of token ids (tensors, arrays or lists) -> strings."""
    out = []
    for seq in _as_lists(seqs):
        words = [vocab.itos[int(i)] for i in seq]
        if remove_special:
            words = [w for w in words if w not in SPECIAL_TOKENS]
        out.append(sep.join(words).replace('@@ ', ''))
    return out


def _ngram_table(tokens, max_order):
    """{n-gram tuple: count} for all orders 1..max_order."""
    table = {}
    n = len(tokens)
    for order in range(1, max_order + 1):
        for start in range(n - order + 1):
            key = tuple(tokens[start:start + order])
            table[key] = table.get(key, 0) + 1
    return table


def corpus_bleu(references, translations, max_order=4, smooth=False):
    """references: per translation a list of tokenised references; translations: tokenised hypotheses.
    Returns (bleu, precisions, brevity_penalty, length_ratio, translation_length, reference_length)."""
    hit = [0] * max_order
    possible = [0] * max_order
    ref_len = hyp_len = 0
    for refs, hyp in zip(references, translations):
        ref_len += min(len(r) for r in refs)
        hyp_len += len(hyp)
        best_ref = {}
        for r in refs:  # per n-gram the largest count over the references
            for key, c in _ngram_table(r, max_order).items():
                if c > best_ref.get(key, 0):
                    best_ref[key] = c
        for key, c in _ngram_table(hyp, max_order).items():
            hit[len(key) - 1] += min(c, best_ref.get(key, 0))
        for order in range(1, max_order + 1):
            possible[order - 1] += max(0, len(hyp) - order + 1)
    if smooth:
        precisions = [(hit[i] + 1.0) / (possible[i] + 1.0) for i in range(max_order)]
    else:
        precisions = [hit[i] / possible[i] if possible[i] > 0 else 0.0 for i in range(max_order)]
    geo_mean = math.exp(sum(math.log(p) for p in precisions) / max_order) if min(precisions) > 0 else 0
    ratio = hyp_len / ref_len if ref_len > 0 else 0.0
    if ratio > 1.0:
        bp = 1.0
    else:
        bp = math.exp(1.0 - 1.0 / ratio) if ratio > 0 else 0.0
    return geo_mean * bp, precisions, bp, ratio, hyp_len, ref_len


def compute_bleu_for_sents(translations, targets, **kwargs):
    """Corpus BLEU of whitespace-tokenised sentence strings against one reference each."""
    return corpus_bleu([[t.split()] for t in targets], [s.split() for s in translations], **kwargs)[0]
