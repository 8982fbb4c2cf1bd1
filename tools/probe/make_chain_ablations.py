"""Builds measurement variants of the row-tile chain kernel (CHAIN_ABL bits: 1 no counted DMA waits, 2 no DMA, 4 no operand
reads, 8 no LDS waits inside the k loops; results are wrong, timings tell what the loop waits for) as libraries of their own
under abl_out/, to be selected with TST_HIP_LIB.   python tools/probe/make_chain_ablations.py 1 3 4 12 15"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'text-style-transfer_amd', 'csrc')
OUT = os.path.join(ROOT, 'abl_out')
os.makedirs(OUT, exist_ok=True)
units = 'capi gemm_f32 gemm_f32_dma gemm_bf16 gru_fused rowops attn attn_mfma select gumbel step'.split()
for a in sys.argv[1:]:
    obj = os.path.join(OUT, 'chain_abl%s.o' % a)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-ffp-contract=off', '-std=c++17', '-DCHAIN_ABL=' + a,
                           '-c', os.path.join(CSRC, 'chain_f32.hip'), '-o', obj])
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(OUT, 'libtst_abl%s.so' % a), obj] +
                          [os.path.join(CSRC, u + '.o') for u in units])
    print('built', a)
