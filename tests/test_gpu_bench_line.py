"""GPU: the one JSON line of ``python bench.py`` (one GPU, a short run with a small CPU sample) carries the fields its readers
depend on, and its figures are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_line_has_the_contract_fields_and_consistent_figures():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--cpu-sample', '16'],
                       capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]                      # exactly one JSON line
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['metric'].startswith('decoded tokens/sec (beam=4, 32k BPE)') and d['unit'] == 'tokens/s'
    assert (d['n_gpus'], d['steps'], d['warmup'], d['higher_is_better'], d['scaling'], d['vs_baseline'], d['dtype'], d['data']) == \
        (1, 2, 1, True, 'weak', None, 'f32', 'synthetic')
    cfg = d['config']
    assert 'workload' in cfg and 'BASELINE config 2' in cfg['workload'] and 'model' not in cfg
    assert cfg['sentences_per_step'] == 8192 and cfg['beam'] == 4 and cfg['vocab'] == 32000
    # value = emitted tokens / time: reproducible from the line's own fields
    emitted_per_step = cfg['mean_emitted_len'] * cfg['sentences_per_step']
    assert abs(d['value'] - emitted_per_step / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    assert 100.0 < d['ms_per_step'] < 400.0
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 157.3
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0.5 < r['frac'] < 1.0
    assert r['traffic'] is None or r['traffic'] > 2.0e9            # at least the logits of an average launch
    # the projection's launches of one batch fit inside the batch time
    assert r['launches'] * r['avg_launch_ms'] < d['ms_per_step']
    s = d['roofline_select']
    assert s['bound'] == 'hbm' and s['unit'] == 'GB/s' and 0.3 < s['frac'] < 1.0
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c and c['unit'] == 'tokens/s'
    assert d['value'] > 50 * c['value']
