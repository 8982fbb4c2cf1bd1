"""Times tst_ln_vocab_argmax_f32 against the three launches it replaces (tst_layernorm_f32, tst_gemm_mfma_f32,
tst_argmax_rows_f32) at the shapes of a small greedy batch.  python tools/bench_vocab_argmax.py [V] [d]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tst_amd  # noqa: E402
from tst_amd import kernels as K  # noqa: E402


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000.0


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    g = torch.Generator().manual_seed(1)
    W = (torch.randn((d, V), generator=g) / 16).cuda()
    bias = torch.randn(V, generator=g).cuda()
    gam, bet = torch.randn(d, generator=g).cuda(), torch.randn(d, generator=g).cuda()
    for R in (256, 192, 128, 64, 16):
        x = torch.randn((R, d), generator=g).cuda()
        h = torch.empty_like(x)
        logits = torch.empty((R, V), dtype=torch.float32, device='cuda')
        tok = torch.empty(R, dtype=torch.int32, device='cuda')
        keys = torch.zeros((R, K.ARGMAX_SLOTS), dtype=torch.int64, device='cuda')

        def separate():
            K.layernorm(x, gam, bet, h, R)
            K.gemm(h, W, bias, logits, R)
            K.argmax_rows(logits, R, V, tok)

        def fused():
            keys.zero_()
            K.ln_vocab_argmax(x, gam, bet, W, bias, keys, R)

        def clear_only():
            keys.zero_()

        t_sep, t_fused, t_clear = timed(separate), timed(fused), timed(clear_only)
        separate(); fused()
        same = bool((K.tokens_of_keys(keys) == tok).all())
        print(json.dumps(dict(rows=R, V=V, d=d, separate_us=round(t_sep, 2), fused_us=round(t_fused - t_clear, 2), clear_us=round(t_clear, 2),
                              same_tokens=same, lib=os.environ.get('TST_HIP_LIB', 'default'))))


if __name__ == '__main__':
    main()
