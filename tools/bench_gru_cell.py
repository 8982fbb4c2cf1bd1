"""Times the GRU cell step at the cyclegan sizes (E = H = 1024): the one-kernel form (tst_gru_cell_fused_f32) against the
three-call form (two products + gate kernel) over a range of row counts, and both with the input pre-activations read from a
per-token table (tst_gru_cell_fused_tab_f32; recurrent product + tst_gru_cell_step_tab_f32).  One JSON line per row count.
Algorithmic work per row-step: 6 * (E * H + H * H) * 2 / 2 = 12.6 MFLOP; TFLOP/s are priced against the fp32 matrix peak (157.3)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tst_amd import kernels as K

E = H = 1024
PEAK = 157.3


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    g = torch.Generator(device='cuda').manual_seed(0)
    wih = torch.randn((E, 3 * H), device='cuda', generator=g) / 32
    whh = torch.randn((H, 3 * H), device='cuda', generator=g) / 32
    bih, bhh = torch.randn(3 * H, device='cuda', generator=g), torch.randn(3 * H, device='cuda', generator=g)
    for R in (128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
        x, h = torch.randn((R, E), device='cuda', generator=g), torch.randn((R, H), device='cuda', generator=g)
        out = torch.empty_like(h)
        gi, gh = torch.empty((R, 3 * H), device='cuda'), torch.empty((R, 3 * H), device='cuda')

        def three():
            K.gemm(x, wih, bih, gi, R)
            K.gemm(h, whh, bhh, gh, R)
            K.gru_cell_step(gi, gh, h, out, R)
        t3 = timeit(three)
        tf = timeit(lambda: K.gru_cell_fused(x, h, wih, bih, whh, bhh, out, R))
        flops = 2.0 * R * 3 * H * (E + H)
        # the same step with a table of V = 128 entries (character vocabulary) in place of the input product
        V = 128
        tab = torch.randn((V, 3 * H), device='cuda', generator=g)
        tok = torch.randint(0, V, (R,), device='cuda', generator=g, dtype=torch.int32)

        def two_tab():
            K.gemm(h, whh, bhh, gh, R)
            K.gru_cell_step_tab(tab, tok, gh, h, out, R)
        t2t = timeit(two_tab)
        tft = timeit(lambda: K.gru_cell_fused_tab(tab, tok, h, whh, bhh, out, R))
        flops_tab = 2.0 * R * 3 * H * H
        print(json.dumps({'rows': R, 'three_call_us': round(t3, 1), 'fused_us': round(tf, 1), 'fused_TFLOPs': round(flops / tf / 1e6, 1),
                          'three_call_TFLOPs': round(flops / t3 / 1e6, 1), 'fused_frac_of_fp32_matrix_peak': round(flops / tf / 1e6 / PEAK, 3),
                          'table_two_call_us': round(t2t, 1), 'table_fused_us': round(tft, 1),
                          'table_fused_TFLOPs': round(flops_tab / tft / 1e6, 1),
                          'table_fused_frac_of_fp32_matrix_peak': round(flops_tab / tft / 1e6 / PEAK, 3)}), flush=True)


if __name__ == '__main__':
    main()
