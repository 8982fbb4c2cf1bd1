"""GPU: InferenceState called with the input variants a caller of the reference might pass (CPU tensors, float64 / float16,
non-contiguous views, mask layouts and types, inputs that require grad, numpy integers as options, a vocabulary object that is
not this package's, a side stream, a state used twice): each gives the outputs of the plain call or a clean Python exception.
The cases live in tools/probe/usage_variants.py and run in a process of their own."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_input_variants_give_the_plain_outputs_or_clean_errors():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'probe', 'usage_variants.py')], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert 'RESULT all ok' in p.stdout, p.stdout[-3000:]
    assert p.stdout.count(' ok\n') >= 18
