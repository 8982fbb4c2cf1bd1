"""Builds ablated variants of the fp32 GEMM kernel (from the product source, by text substitution) next to
this file as libgemm_<variant>.so; bench_variants.py times them (results of the ablated variants are wrong on
purpose).  Variants: BASE; NOLDSREAD (operands not read from LDS); WIDEREAD (a quarter of the operand read
instructions, each 16 bytes wide: the LDS instruction count of a k-permuted operand image at about the same bytes).
Earlier measurements with this tool on the vocabulary projection 32768 x 32000 x 256: BASE 120 TF, no barrier 121,
no epilogue stores 120, no staging 135, NOLDSREAD 137 TFLOP/s."""
import os
import subprocess

here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, '..', '..', 'text-style-transfer_amd', 'csrc')
s = open(os.path.join(csrc, 'gemm_f32.hip')).read()


def sub(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)


sub('#include "tst_common.h"', '#include "../../text-style-transfer_amd/csrc/tst_common.h"')
sub("""#define TST_READ_OPERANDS(av, bv, bufi, kk)                                                                   \\
    _Pragma("unroll") for (int i = 0; i < WTM; ++i) av[i] = As[bufi][(kk) + khalf][arow + 32 * i];            \\
    _Pragma("unroll") for (int j = 0; j < WTN; ++j) bv[j] = Bs[bufi][(kk) + khalf][bcol + 32 * j];""",
    """#if defined(ABL_NOLDSREAD)
#define TST_READ_OPERANDS(av, bv, bufi, kk)                                                                   \\
    _Pragma("unroll") for (int i = 0; i < WTM; ++i) av[i] = av[i] * 1.0001f;                                  \\
    _Pragma("unroll") for (int j = 0; j < WTN; ++j) bv[j] = bv[j] * 1.0001f;
#elif defined(ABL_WIDEREAD)
#define TST_READ_OPERANDS(av, bv, bufi, kk)                                                                   \\
    if (((kk) & 7) == 0) {                                                                                    \\
        _Pragma("unroll") for (int i = 0; i < WTM; ++i) wa[i] = *reinterpret_cast<const float4*>(&As[bufi][(kk) + khalf][(arow & ~3) + 32 * i]); \\
        _Pragma("unroll") for (int j = 0; j < WTN; ++j) wb[j] = *reinterpret_cast<const float4*>(&Bs[bufi][(kk) + khalf][(bcol & ~3) + 32 * j]); \\
    }                                                                                                         \\
    _Pragma("unroll") for (int i = 0; i < WTM; ++i) av[i] = ((kk) & 6) == 0 ? wa[i].x : ((kk) & 6) == 2 ? wa[i].y : ((kk) & 6) == 4 ? wa[i].z : wa[i].w; \\
    _Pragma("unroll") for (int j = 0; j < WTN; ++j) bv[j] = ((kk) & 6) == 0 ? wb[j].x : ((kk) & 6) == 2 ? wb[j].y : ((kk) & 6) == 4 ? wb[j].z : wb[j].w;
#else
#define TST_READ_OPERANDS(av, bv, bufi, kk)                                                                   \\
    _Pragma("unroll") for (int i = 0; i < WTM; ++i) av[i] = As[bufi][(kk) + khalf][arow + 32 * i];            \\
    _Pragma("unroll") for (int j = 0; j < WTN; ++j) bv[j] = Bs[bufi][(kk) + khalf][bcol + 32 * j];
#endif""")
sub("    float a_cur[WTM], b_cur[WTN], a_nxt[WTM], b_nxt[WTN];",
    "    float a_cur[WTM], b_cur[WTN], a_nxt[WTM], b_nxt[WTN];\n    float4 wa[WTM], wb[WTN];\n"
    "    for (int i = 0; i < WTM; ++i) { a_nxt[i] = 1.f + lane; a_cur[i] = 2.f; wa[i] = make_float4(1.f, 2.f, 3.f, 4.f); }\n"
    "    for (int j = 0; j < WTN; ++j) { b_nxt[j] = 1.f; b_cur[j] = 0.5f * lane; wb[j] = make_float4(1.f, 2.f, 3.f, 4.f); }")
src = os.path.join(here, '_gemm_ablate.hip')
open(src, 'w').write(s)
for v in ('BASE', 'NOLDSREAD', 'WIDEREAD'):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-ffp-contract=off', '-std=c++17',
                           '-DABL_' + v, '-shared', '-o', os.path.join(here, 'libgemm_%s.so' % v), src,
                           os.path.join(csrc, 'capi.hip')])
os.remove(src)
