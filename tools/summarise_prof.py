"""Summarises rocprofv3 output (the default results.db) of a bench.py run into the text files kept under profiles/.

  python tools/summarise_prof.py stats <dir>            -> kernel table (calls, total ms, average us, share)
  python tools/summarise_prof.py pmc   <dir> <counter>  -> per-kernel sum and per-dispatch average of a counter

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


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
