"""Builds ablated variants of the bf16 GEMM (from the product source, by text substitution) next to this file as
libbf16_<variant>.so; bench_bf16_variants.py times them on the vocabulary projection (ablated results are wrong on
purpose).  BASE; NOSTORE (the epilogue runs, its global stores do not); NOMAIN (no k loop: prologue + epilogue only)."""
import os
import subprocess

here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, '..', '..', 'text-style-transfer_amd', 'csrc')
s = open(os.path.join(csrc, 'gemm_bf16.hip')).read()


def sub(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)


sub('#include "tst_common.h"', '#include "../../text-style-transfer_amd/csrc/tst_common.h"')
sub("""                if (grow >= M || gcol >= N) continue;
                const uint4 o = *reinterpret_cast<const uint4*>(img + lrow * OUT_LD + c8 * 2);""",
    """#if defined(ABL_NOSTORE)
                if (grow >= M || gcol >= N || tiles_m > 0) continue;
#else
                if (grow >= M || gcol >= N) continue;
#endif
                const uint4 o = *reinterpret_cast<const uint4*>(img + lrow * OUT_LD + c8 * 2);""")
sub("    const int nkt = K / DK;\n    // fragment read", """#if defined(ABL_NOMAIN)
    const int nkt = tiles_m > 0 ? 0 : K / DK;
#else
    const int nkt = K / DK;
#endif
    // fragment read""")
sub("""    const int per_group = GROUP_M * tiles_n;
    const int first_m = (lb / per_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (lb % per_group) % gsz;
    const int tn = (lb % per_group) / gsz;
    const int m0 = tm * TM, n0 = tn * TN;""", """#if defined(ABL_G8)
    constexpr int GM = 8;
#elif defined(ABL_G4)
    constexpr int GM = 4;
#elif defined(ABL_G2)
    constexpr int GM = 2;
#else
    constexpr int GM = GROUP_M;
#endif
    const int per_group = GM * tiles_n;
    const int first_m = (lb / per_group) * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (lb % per_group) % gsz;
    const int tn = (lb % per_group) / gsz;
    const int m0 = tm * TM, n0 = tn * TN;""")
sub("""                if (gcol + 8 <= N) {
                    *reinterpret_cast<uint4*>(cp) = o;
                } else {
                    Chunk oc;
                    oc.u = o;""", """                if (gcol + 8 <= N) {
#if defined(ABL_NT)
                    u32x4 ov; ov[0] = o.x; ov[1] = o.y; ov[2] = o.z; ov[3] = o.w;
                    __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(cp));
#else
                    *reinterpret_cast<uint4*>(cp) = o;
#endif
                } else {
                    Chunk oc;
                    oc.u = o;""")
sub("gemm_bf16_dma_kernel<2, 4, 4, 2, 3, OB, HR, RL>", """gemm_bf16_dma_kernel<2, 4, 4, 2, ABL_NST, OB, HR, RL>""")
s = "#ifndef ABL_NST\n#define ABL_NST 3\n#endif\n" + s
sub("""        if (kt + NST < nkt) stage(st, (kt + NST) * DK);
        if (kt + 1 < nkt) read_step(st_next, 0);""", """#if defined(ABL_NODMA)
        if (kt + NST < nkt && tiles_m < 0) stage(st, (kt + NST) * DK);
#else
        if (kt + NST < nkt) stage(st, (kt + NST) * DK);
#endif
#if defined(ABL_NOREAD)
        if (kt + 1 < nkt && tiles_m < 0) read_step(st_next, 0);
#else
        if (kt + 1 < nkt) read_step(st_next, 0);
#endif""")
sub("""        lds_wait();
        read_step(st, 1);
        mfma_step(0);""", """        lds_wait();
#if defined(ABL_NOREAD)
        if (tiles_m < 0) read_step(st, 1);
#else
        read_step(st, 1);
#endif
        mfma_step(0);""")
sub("""        lds_wait();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#if defined(ABL_NODMA)""", """        lds_wait();
#if !defined(ABL_NOBAR)
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
#if defined(ABL_NODMA)""")
src = os.path.join(here, '_bf16_ablate.hip')
open(src, 'w').write(s)
import sys
for v in (sys.argv[1:] or ['BASE', 'NOSTORE', 'NOMAIN', 'G8', 'G4', 'G2']):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-ffp-contract=off', '-std=c++17']
                          + ['-DABL_' + x for x in v.split('_')] + (['-DABL_NST=4'] if 'S4' in v.split('_') else []) + [ '-shared', '-o', os.path.join(here, 'libbf16_%s.so' % v), src,
                           os.path.join(csrc, 'capi.hip')])
os.remove(src)
