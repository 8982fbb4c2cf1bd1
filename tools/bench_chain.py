"""GPU: the row-tile chain kernel against the three launches it replaces (residual product, LayerNorm, product) at the
row counts of a config-2 decode step.  Prints microseconds per chain and the TFLOP/s of the chain kernel.
    python tools/bench_chain.py [rows ...]"""
import sys
import torch
sys.path.insert(0, '.')
import tst_amd  # noqa: F401
from tst_amd import kernels as K

CHAINS = [('o -> ln -> cross-q', 256, 256, 0), ('cross-o -> ln -> ff1', 256, 1024, 1), ('ff2 -> ln -> q|k|v', 1024, 768, 0),
          ('ff2 -> ln (k-major)', 1024, 0, 0)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    rows = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192, 12288, 15680, 16384, 20480, 24576, 32768]
    g = torch.Generator(device='cuda').manual_seed(1)
    for R in rows:
        for name, K1, N2, act in CHAINS:
            a = torch.randn((R, K1), device='cuda', generator=g)
            w1 = torch.randn((K1, 256), device='cuda', generator=g) / K1 ** 0.5
            b1 = torch.randn(256, device='cuda', generator=g)
            x = torch.randn((R, 256), device='cuda', generator=g)
            gam, bet = torch.randn(256, device='cuda', generator=g), torch.randn(256, device='cuda', generator=g)
            w2 = torch.randn((256, max(N2, 4)), device='cuda', generator=g) / 16
            b2 = torch.randn(max(N2, 4), device='cuda', generator=g)
            h = torch.empty((R, 256), device='cuda')
            out = torch.empty((R, max(N2, 4)), device='cuda')
            ldt = (R + 127) // 128 * 128
            out_t = torch.empty((256, ldt), device='cuda')

            def separate():
                K.gemm(a, w1, b1, x, R, res=x)
                if N2:
                    K.layernorm(x, gam, bet, h, R)
                    K.gemm(h, w2, b2, out, R, n_cols=N2, act=act)
                else:
                    K.layernorm_t(x, gam, bet, out_t, R)

            def chain():
                if N2:
                    K.chain(a, w1, b1, x, gam, bet, R, w2t=w2, b2=b2, out=out, n_cols=N2, act2=act)
                else:
                    K.chain(a, w1, b1, x, gam, bet, R, out_t=out_t)

            ts, tc = timed(separate), timed(chain)
            flop = 2.0 * R * 256 * (K1 + N2)
            print('rows %6d  %-22s separate %7.1f us   chain %7.1f us   (%5.1f TF)   %+5.1f %%' % (R, name, ts, tc, flop / tc / 1e6,
                                                                                              (ts / tc - 1) * 100), flush=True)


if __name__ == '__main__':
    main()
