"""Every one-off GPU run script under tools/runs/ has its line (question -> answer) in tools/runs/README."""
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_run_script_is_indexed():
    readme = open(os.path.join(ROOT, 'tools', 'runs', 'README')).read()
    indexed = {line.split()[0] for line in readme.splitlines() if line[:1] == 'r' and '_' in line.split()[0]}
    scripts = {os.path.basename(p)[:-3] for p in glob.glob(os.path.join(ROOT, 'tools', 'runs', '*.sh'))}
    assert scripts <= indexed, sorted(scripts - indexed)
