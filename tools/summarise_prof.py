"""Summarises rocprofv3 output (the default results.db) of a bench.py run into the text files kept under profiles/.

  python tools/summarise_prof.py stats <dir>            -> kernel table (calls, total ms, average us, share)
  python tools/summarise_prof.py pmc   <dir> <counter>  -> per-kernel sum and per-dispatch average of a counter
  python tools/summarise_prof.py util  <sq dir> <fetch dir> <write dir>
                                                        -> per kernel: matrix-pipe busy share, clock, HBM-side GB/s

GEMM dispatches are split into the vocabulary projection (N = 32000: the workgroup count is a multiple of
250 column tiles) and the layer GEMMs.  FETCH_SIZE and WRITE_SIZE are in units of 1024 bytes; on gfx950
FETCH_SIZE counts 128-byte read requests of a wide streaming read as 64 bytes, so the bytes read are about
twice the reported value for such kernels.
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def open_db(d):
    hits = glob.glob(os.path.join(d, '**', '*results.db'), recursive=True)
    if not hits:
        raise SystemExit('no results.db under %s' % d)
    return sqlite3.connect(max(hits, key=os.path.getsize))   # a run with child processes leaves one database per process: the bench's is the largest


def label(name, grid, wg):
    key = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:90]
    if 'gemm_f32_dma_kernel' in key:      # only the vocabulary projection uses the LDS-DMA kernel
        key += ' [vocabulary projection]'
    elif 'gemm_' in key:
        n = grid // max(1, wg)
        cols = 125 if '<2, 4,' in key else 250          # column tiles of N = 32000: 256- or 128-wide tiles
        key += ' [vocabulary projection]' if n >= cols and n % cols == 0 else ' [layer GEMMs]'
    return key


def stats(d):
    rows = open_db(d).execute('select name, grid_x, workgroup_x, duration from kernels').fetchall()
    agg = defaultdict(lambda: [0, 0.0])
    for name, grid, wg, dur in rows:
        a = agg[label(name, grid, wg)]
        a[0] += 1
        a[1] += dur * 1e-3
    total = sum(v[1] for v in agg.values())
    print('%-100s %8s %12s %10s %7s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'share'))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-100s %8d %12.3f %10.1f %6.1f%%' % (k, n, t * 1e-3, t / n, 100 * t / total))
    print('total GPU kernel time %.1f ms over %d dispatches' % (total * 1e-3, len(rows)))


def pmc(d, counter):
    rows = open_db(d).execute('select kernel_name, grid_size, workgroup_size, value, duration from counters_collection '
                              'where counter_name = ?', (counter,)).fetchall()
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for name, grid, wg, val, dur in rows:
        a = agg[label(name, grid, wg)]
        a[0] += 1
        a[1] += float(val)
        a[2] += dur * 1e-3
    print('%-100s %8s %16s %16s %10s' % ('kernel', 'calls', counter + ' sum', 'per dispatch', 'avg_us'))
    for k, (n, t, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-100s %8d %16.6g %16.6g %10.1f' % (k, n, t, t / n, us / n))


def per_kernel(d, counter):
    rows = open_db(d).execute('select kernel_name, grid_size, workgroup_size, value, duration from counters_collection '
                              'where counter_name = ?', (counter,)).fetchall()
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for name, grid, wg, val, dur in rows:
        a = agg[label(name, grid, wg)]
        a[0] += 1
        a[1] += float(val)
        a[2] += dur * 1e-9
    return agg


def util(sq, fetch, write):
    """Matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs), cycles = GRBM_GUI_ACTIVE / 8 XCDs (which also gives the
    clock the kernel ran at); fabric bytes = 2 x FETCH_SIZE + WRITE_SIZE KiB (the gfx950 correction for 16-byte-per-lane reads:
    an upper bound for kernels that read narrower), over the kernel's time in that pass."""
    mf, gui = per_kernel(sq, 'SQ_VALU_MFMA_BUSY_CYCLES'), per_kernel(sq, 'GRBM_GUI_ACTIVE')
    fe, wr = per_kernel(fetch, 'FETCH_SIZE'), per_kernel(write, 'WRITE_SIZE')
    print('# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); GHz = GRBM_GUI_ACTIVE / 8 / kernel time.  GRBM_GUI_ACTIVE')
    print('# also counts the dispatch gaps around a kernel: for kernels under ~100 us the GHz column reads high (the chip does not exceed 2.4)')
    print('# and the busy share correspondingly low (scale it by GHz / 2.3).  read = 2 x FETCH_SIZE (an upper bound for narrow reads), from')
    print('# the fabric side of the L2 (Infinity Cache hits included); each counter comes from a pass of its own over the same command.')
    print('%-92s %7s %9s %7s %9s %10s %10s' % ('kernel', 'calls', 'avg_us', 'GHz', 'MFMA busy', 'read GB/s', 'write GB/s'))
    for k, (n, g, t) in sorted(gui.items(), key=lambda kv: -kv[1][2]):
        cyc = g / 8.0
        busy = 100.0 * mf[k][1] / (cyc * 1024) if k in mf and cyc > 0 else 0.0
        rd = 2 * fe[k][1] * 1024 / fe[k][2] / 1e9 if k in fe and fe[k][2] > 0 else 0.0
        ww = wr[k][1] * 1024 / wr[k][2] / 1e9 if k in wr and wr[k][2] > 0 else 0.0
        print('%-92s %7d %9.1f %7.2f %8.1f%% %10.0f %10.0f' % (k[:92], n, t / n * 1e6, cyc / t / 1e9, busy, rd, ww))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    elif sys.argv[1] == 'util':
        util(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3])
