"""Builds variants of the LDS-DMA vocabulary GEMM (text substitution on the product source) as libvdma_<name>.so next to this
file: m-group sizes of the tile order and (k per stage, ring depth) pairs; bench_vdma_variants.py times them."""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, '..', '..', 'text-style-transfer_amd', 'csrc')
src0 = open(os.path.join(csrc, 'gemm_f32_dma.hip')).read()
variants = {'gm16_8x6': (16, 8, 6), 'gm8_8x6': (8, 8, 6), 'gm4_8x6': (4, 8, 6), 'gm32_8x6': (32, 8, 6), 'gm16_8x8': (16, 8, 8),
            'gm16_8x5': (16, 8, 5), 'gm16_16x4': (16, 16, 4), 'gm8_16x4': (8, 16, 4)}
for name in (sys.argv[1:] or variants):
    gm, bk, nst = variants[name]
    s = src0.replace('#include "tst_common.h"', '#include "../../text-style-transfer_amd/csrc/tst_common.h"')
    assert 'constexpr int TN = 256, GROUP_M = 16;' in s
    s = s.replace('constexpr int TN = 256, GROUP_M = 16;', 'constexpr int TN = 256, GROUP_M = %d;' % gm)
    assert 'case 3: launch_plain128<8, 6, true>' in s
    s = s.replace('case 3: launch_plain128<8, 6, true>', 'case 3: launch_plain128<%d, %d, true>' % (bk, nst))
    p = os.path.join(here, '_vdma.hip')
    open(p, 'w').write(s)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-ffp-contract=off', '-std=c++17', '-shared', '-o',
                           os.path.join(here, 'libvdma_%s.so' % name), p, os.path.join(csrc, 'capi.hip')])
    os.remove(p)
