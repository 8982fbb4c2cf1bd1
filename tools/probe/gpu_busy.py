"""Share of the time between the first and the last kernel of the LAST batch in which a kernel was running
(rocprofv3 results.db of tools/probe/cfg1_once.py): python tools/probe/gpu_busy.py <dir> <dispatches per batch>"""
import glob
import os
import sqlite3
import sys

db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], '**', '*results.db'), recursive=True)[0])
rows = db.execute('select start, end from kernels order by start').fetchall()
n = len(rows) // 3
last = rows[-n:]
span = last[-1][1] - last[0][0]
busy = sum(e - s for s, e in last)
gaps = sorted((last[i + 1][0] - last[i][1]) for i in range(len(last) - 1))
print('last third of the trace: %d dispatches over %.2f ms, kernels running %.2f ms (%.0f %%); gap median %.1f us, p90 %.1f us, max %.1f us'
      % (len(last), span / 1e6, busy / 1e6, 100.0 * busy / span, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3))
