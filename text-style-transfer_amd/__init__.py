"""MI355X-native batched decoding (greedy / beam) for transformer and GRU encoder-decoders.

Host side of ``libtst_hip.so``: the call surface of the reference's ``src/inference.py`` and of the model
classes it drives, with all per-step arithmetic in hand-written gfx950 kernels.  Importing this package
requires the built library (``csrc/build.sh``); there is no CPU or PyTorch fallback path.
"""
from . import _lib
from ._lib import TstError, version
from .vocab import Vocab
from .inference import InferenceState, simple_inference
from .transformer import HP, Transformer, TransformerDecoder
from .encoder import TransformerEncoder
from .rnn import CharLMFromEmbs, ConditionalLM, FFN, Generator, RNNDecoder, RNNEncoder, RNNLM, WeightedLMEnsemble
from .lm import TransformerLM
from .postprocess import itos_many, compute_bleu_for_sents, corpus_bleu
from .streams import run_on_streams

__all__ = ['InferenceState', 'simple_inference', 'Vocab', 'HP', 'Transformer', 'TransformerDecoder', 'TransformerEncoder',
           'RNNDecoder', 'RNNEncoder', 'Generator', 'FFN', 'RNNLM', 'ConditionalLM', 'WeightedLMEnsemble', 'CharLMFromEmbs',
           'TransformerLM', 'itos_many', 'compute_bleu_for_sents', 'corpus_bleu', 'run_on_streams', 'TstError', 'version']
