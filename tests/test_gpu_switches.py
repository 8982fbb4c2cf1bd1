"""GPU: the environment switches that route a decode step through alternative kernels (the row-tile chain kernels, the
persistent vocabulary projection, the LayerNorm epilogue of the bf16 residual products, the K = 256 products with the weights
held in registers) are read once per process, so each is
exercised in a child process: the same synthetic batches must come out token for token as in the default process (whose
vocabulary projection at d_model 256 is the weights-in-registers kernel; TST_VOCAB_VARIANT=3 is the LDS-DMA kernel)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, json, sys, torch
sys.path.insert(0, %r)
import tst_amd
from tst_amd import synthetic
out = {}
# transformer-mt sizes, 2304 sentences x beam 4 = 9216 rows in the first steps: chain kernels and the LDS-DMA projection apply
hp = dict(d_model=256, dropout=0.1, n_layers=2, pe_max_len=60, n_heads=2, d_ff=1024)
V = 4000
vocab = tst_amd.Vocab.synthetic(V)
model = tst_amd.Transformer(tst_amd.HP(hp), vocab, vocab)
model.load_state_dict(synthetic.synthetic_state_dict(model, 3, 4.0, 2.0, seed=7))
src = synthetic.multi30k_like_sources(2304, V, seed=99).cuda()
encs, mask = model.encoder(src)
st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 24, 'beam_size': 4})
tok, lens, score = st.decode_packed()
out['mt'] = hashlib.sha256(tok.cpu().numpy().tobytes() + lens.cpu().numpy().tobytes() + score.cpu().numpy().tobytes()).hexdigest()
del st, model
# z-conditioned bf16 language model, model_size 512: 600 sentences x beam 8 = 4800 rows
hp = dict(model_size=512, n_layers=2, n_heads=8, d_ff=2048, pe_max_len=32)
lm = tst_amd.TransformerLM(tst_amd.HP(hp), vocab, dtype='bf16')
lm.load_state_dict(synthetic.synthetic_state_dict(lm, 3, 4.0, 2.0, seed=8))
z = (0.3 * torch.randn((600, 512), generator=torch.Generator().manual_seed(5))).cuda()
st = tst_amd.InferenceState({'model': lm, 'inputs': z, 'vocab': vocab, 'max_len': 16, 'beam_size': 8})
tok, lens, score = st.decode_packed()
out['lm'] = hashlib.sha256(tok.cpu().numpy().tobytes() + lens.cpu().numpy().tobytes() + score.cpu().numpy().tobytes()).hexdigest()
# the K = 256 layer products at row counts beyond the one-wave strips, with bias / relu / residual, rows no multiple of the tile
from tst_amd import kernels as K
g = torch.Generator(device='cuda').manual_seed(11)
hsh = hashlib.sha256()
for M, N, res, act in ((3300, 256, True, 0), (5001, 768, False, 0), (4100, 1024, False, 1), (20481, 256, True, 1), (9000, 512, False, 0), (4000, 300, True, 0), (3500, 40, False, 1)):
    a = torch.randn((M, 256), device='cuda', generator=g)
    w = torch.randn((256, N), device='cuda', generator=g) / 16
    b = torch.randn(N, device='cuda', generator=g)
    r = torch.randn((M, N), device='cuda', generator=g) if res else None
    o = torch.full((M + 1, N), 7.0, device='cuda')
    K.gemm(a, w, b, o, M, res=r, act=act)
    assert (o[M] == 7.0).all()
    hsh.update(o[:M].cpu().numpy().tobytes())
    if res:                                         # in place: the output is the residual
        K.gemm(a, w, b, r, M, res=r, act=act)
        assert torch.equal(r, o[:M])
out['gemm_k256'] = hsh.hexdigest()
print('RESULT ' + json.dumps(out))
''' % ROOT


def run_child(extra_env):
    env = dict(os.environ)
    for k in ('TST_TFM_CHAIN', 'TST_VOCAB_VARIANT', 'TST_BF16_LN_FUSE', 'TST_GEMM_WREG', 'TST_LN_WREG', 'TST_WREG_ROW_BLOCKS'):
        env.pop(k, None)
    env.update(extra_env)
    p = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1]
    return json.loads(line[len('RESULT '):])


@pytest.fixture(scope='module')
def default_result():
    assert torch.cuda.is_available()
    return run_child({})


@pytest.mark.parametrize('env', [{'TST_TFM_CHAIN': '1'}, {'TST_VOCAB_VARIANT': '4'}, {'TST_VOCAB_VARIANT': '2'},
                                 {'TST_BF16_LN_FUSE': '256'}, {'TST_TFM_CHAIN': '1', 'TST_VOCAB_VARIANT': '4'}, {'TST_GEMM_WREG': '1'}, {'TST_VOCAB_VARIANT': '3'},
                                 {'TST_VOCAB_VARIANT': '3', 'TST_GEMM_WREG': '1'}, {'TST_LN_WREG': '1'}, {'TST_LN_WREG': '2'},
                                 {'TST_WREG_ROW_BLOCKS': '1'}])
def test_alternative_kernels_give_the_default_outputs(default_result, env):
    assert run_child(env) == default_result
