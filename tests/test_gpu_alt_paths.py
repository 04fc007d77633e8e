"""The library keeps alternative launch paths behind environment switches (tools/README.md): per-layer kernels instead of the fused chain
kernels (MVSDF_FUSE=0), the backward pass as separate E.1 / E.2 chain launches (MVSDF_SPLIT_CHAINS=1), 8-wave chain workgroups
(MVSDF_CHAIN_W8=1), two row tiles per chain workgroup everywhere (MVSDF_CHAIN_MT=2).  Each is an independent implementation of the same
passes: the reference-fixture tests of the differentiable kernels and one end-to-end fixture must pass on every one of them.  The switches
are read once per process, so each configuration runs in a child pytest."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

TARGETS = ['tests/test_gpu_diff.py', 'tests/test_gpu_idr.py::test_forward_loss_backward_vs_reference[idr_w64_tp03]',
           'tests/test_gpu_idr.py::test_forward_loss_backward_vs_reference[idr_w256_tp03]']


@pytest.mark.parametrize('env', [{'MVSDF_FUSE': '0'}, {'MVSDF_SPLIT_CHAINS': '1'}, {'MVSDF_CHAIN_W8': '1'}, {'MVSDF_CHAIN_MT': '2'}],
                         ids=lambda e: ','.join('%s=%s' % kv for kv in e.items()))
def test_reference_fixtures_pass_on_the_alternative_paths(env):
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider'] + TARGETS, cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = p.stdout.decode(errors='replace')[-2500:]
    assert p.returncode == 0, tail
    assert ' passed' in tail and 'failed' not in tail, tail
