"""Builds ablated variants of the fp32 GEMM (from the product source, by text substitution) next to this file as
liblayer_<variant>.so for the layer products; bench_layer_variants.py times them (ablated results are wrong on purpose).
BASE; NOLOAD (no global loads inside the k loop: the staged registers keep the second tile's values); NOBAR (no barrier in the loop)."""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, '..', '..', 'text-style-transfer_amd', 'csrc')
s = open(os.path.join(csrc, 'gemm_f32.hip')).read()


def sub(old, new, count=1):
    global s
    assert s.count(old) >= count, old
    s = s.replace(old, new)


sub('#include "tst_common.h"', '#include "../../text-style-transfer_amd/csrc/tst_common.h"')
sub("""                    if (kt + 2 < nkt) TST_LOAD_TILE((kt + 2) * BK);
                }
                __syncthreads();""", """#if !defined(ABL_NOLOAD)
                    if (kt + 2 < nkt) TST_LOAD_TILE((kt + 2) * BK);
#endif
                }
#if !defined(ABL_NOBAR)
                __syncthreads();
#endif""")
sub("""                    if (kt + 2 < nkt) load_tile((kt + 2) * KT);
                }
                __syncthreads();""", """#if !defined(ABL_NOLOAD)
                    if (kt + 2 < nkt) load_tile((kt + 2) * KT);
#endif
                }
#if !defined(ABL_NOBAR)
                __syncthreads();
#endif""")
src = os.path.join(here, '_layer_ablate.hip')
open(src, 'w').write(s)
for v in (sys.argv[1:] or ['BASE', 'NOLOAD', 'NOBAR', 'NOLOAD_NOBAR']):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-ffp-contract=off', '-std=c++17']
                          + ['-DABL_' + x for x in v.split('_')] + ['-shared', '-o', os.path.join(here, 'liblayer_%s.so' % v), src,
                                                                    os.path.join(csrc, 'capi.hip'), os.path.join(csrc, 'rowops.hip')])
os.remove(src)
