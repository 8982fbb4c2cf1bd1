"""Vocabulary with the behaviour the decode loop relies on (``/root/reference/src/inference.py:23-26,57-58``):
``stoi`` maps unknown strings to the ``<unk>`` id (legacy torchtext ``Vocab.stoi`` is a defaultdict),
``itos`` is the id -> string list, ``len(vocab)`` the vocabulary size.  Any object with these three
members (e.g. a pickled torchtext vocab) works in its place."""


class _Stoi(dict):
    def __init__(self, mapping, unk_id):
        super().__init__(mapping)
        self._unk = unk_id

    def __missing__(self, key):
        return self._unk


class Vocab:
    SPECIALS = ('<unk>', '<pad>', '<bos>', '<eos>')

    def __init__(self, tokens, specials=SPECIALS):
        self.itos = list(specials) + [t for t in tokens if t not in specials]
        self.stoi = _Stoi({t: i for i, t in enumerate(self.itos)}, self.itos.index('<unk>') if '<unk>' in self.itos else 0)

    def __len__(self):
        return len(self.itos)

    @classmethod
    def synthetic(cls, size, extra=()):
        """Specials at ids 0..3, then ``extra``, then filler tokens ``w<i>`` up to ``size`` entries."""
        toks = list(extra)
        toks += ['w%d' % i for i in range(size - len(cls.SPECIALS) - len(toks))]
        v = cls(toks)
        if len(v) != size:
            raise ValueError('cannot build a vocabulary of size %d' % size)
        return v
