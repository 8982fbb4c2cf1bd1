#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_attn_small; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -- python3 tools/bench_configs.py cfg1 --sentences 256 > $out/log 2>&1 || { tail $out/log; exit 1; }
python3 - <<'PY'
import glob, os, sqlite3
hits = glob.glob('gpurun_out/prof_attn_small/trace/**/*results.db', recursive=True)
db = sqlite3.connect(max(hits, key=os.path.getsize))
rows = db.execute("select name, start, duration, grid_x from kernels order by start").fetchall()
# last batch: take the last 50*? dispatches; print self-attention durations in order
sel = [(n, s, d, g) for n, s, d, g in rows if 'attn_step_kernel' in n]
tail = sel[-200:]
print('self (true) durations us, last 100 launches:', [round(d / 1e3, 1) for n, s, d, g in tail if 'true' in n.split('attn_step_kernel')[1][:20]][-100:])
print('cross (false) durations us:', [round(d / 1e3, 1) for n, s, d, g in tail if 'false' in n.split('attn_step_kernel')[1][:20]][-100:])
names = {}
for n, s, d, g in rows[-400:]:
    k = n.split('(')[0][-60:]
    names.setdefault(k, []).append(round(d / 1e3, 1))
for k, v in names.items():
    print(k, len(v), 'min', min(v), 'max', max(v))
# gaps between consecutive kernels in the last 300 dispatches
gaps = [rows[i + 1][1] - (rows[i][1] + rows[i][2]) for i in range(len(rows) - 300, len(rows) - 1)]
gaps.sort()
print('gap us: median %.1f p90 %.1f' % (gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * .9)] / 1e3))
PY
rm -rf $out/trace
