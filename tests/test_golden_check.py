"""Fixture hygiene: the committed .npz files must be what tests/golden/make_golden.py writes TODAY.  Where the PyTorch reference is present (the build container;
it never travels to the GPU box) two small fixtures are regenerated into a scratch directory and compared with the committed files array by array, bit for bit
(`make_golden.py --check`): a fixture that drifted from its generator -- arrays missing, values changed -- fails here."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(not os.path.isdir('/root/reference/code'), reason='the PyTorch reference is not on this machine')
def test_two_fixtures_regenerate_bit_for_bit():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'), '--check', 'sdf_w64', 'idr_eval_w64', 'idr_w64_tp03'],
                       capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert r.stdout.count('identical') == 3
