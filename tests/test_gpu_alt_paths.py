"""The DEVELOPMENT build of the library (mvsdf_amd/build.py: build(tag='dev'), -DMVSDF_DEV_SWITCHES; loaded through MVSDF_LIB) keeps alternative launch paths
behind environment switches the product library does not read (tools/README.md): per-layer kernels instead of the fused chain
kernels (MVSDF_FUSE=0), the backward pass as separate E.1 / E.2 chain launches (MVSDF_SPLIT_CHAINS=1), 8-wave chain workgroups
(MVSDF_CHAIN_W8=1), two row tiles per chain workgroup everywhere (MVSDF_CHAIN_MT=2), the delta pass as a chain of GEMMs instead of the
scaling of the saved s_l (MVSDF_DELTA_CHAIN=1), the Python-orchestrated step (MVSDF_NATIVE_STEP=0), the fused SDF chains on the fp32-input MFMA instead of the three-term bf16 form (MVSDF_CHAIN_X3=0), the tracer without tail filling (MVSDF_TAIL=0), the step's sample rows evaluated on a side
stream beside the tracer at every size (MVSDF_SPLIT_ROWS=1; by default only where the rays' rows alone make a shorter launch), the step's
CPU-generator draws delivered by an async copy in front of the step instead of being read from pinned memory by its first kernel
(IDRNetwork.host_stage = False, set through tests/conftest.py's MVSDF_TEST_HOST_STAGE=0).  Each is an independent implementation of the same
passes: the reference-fixture tests of the differentiable kernels and one end-to-end fixture must pass on every one of them.  The switches
are read once per process, so each configuration runs in a child pytest."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev_lib():
    from mvsdf_amd import build
    return build.build(tag='dev')

# (the end-to-end fixtures in both orders: 'loss_first' = the deferred step where the configuration has one -- the per-layer routes wait for the counts instead)
TARGETS = ['tests/test_gpu_diff.py', 'tests/test_gpu_idr.py::test_forward_loss_backward_vs_reference[idr_w64_tp03-outputs_first]',
           'tests/test_gpu_idr.py::test_forward_loss_backward_vs_reference[idr_w64_tp03-loss_first]',
           'tests/test_gpu_idr.py::test_forward_loss_backward_vs_reference[idr_w256_tp03-loss_first]',
           # a skip connection into the LAST Linear (idr.py:46-49,86): the one layout the per-layer route treats apart (k_pe_adj_top, bcast_sqrt2)
           'tests/test_gpu_options.py::test_several_skip_connections[sdf_bwd_w64_skip8]',
           'tests/test_gpu_idr.py::test_forward_loss_backward_vs_reference[idr_w64_skip8-outputs_first]']


@pytest.mark.parametrize('env', [{'MVSDF_FUSE': '0'}, {'MVSDF_SPLIT_CHAINS': '1'}, {'MVSDF_CHAIN_W8': '1'}, {'MVSDF_CHAIN_MT': '2'}, {'MVSDF_DELTA_CHAIN': '1'},
                                 {'MVSDF_NATIVE_STEP': '0'}, {'MVSDF_CHAIN_X3': '0'}, {'MVSDF_TAIL': '0'}, {'MVSDF_SPLIT_ROWS': '1'}, {'MVSDF_TEST_HOST_STAGE': '0'}],
                         ids=lambda e: ','.join('%s=%s' % kv for kv in e.items()))
def test_reference_fixtures_pass_on_the_alternative_paths(env, dev_lib):
    e = dict(os.environ, MVSDF_LIB=dev_lib)
    e.update(env)
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider'] + TARGETS, cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = p.stdout.decode(errors='replace')[-2500:]
    assert p.returncode == 0, tail
    assert ' passed' in tail and 'failed' not in tail, tail


def test_delta_by_scaling_equals_the_delta_chain(dev_lib):
    """SampleNetwork's scalar enters the SDF backward as one extra upstream per hit row on output column 0.  The library adds its adjoints as
    fbar x s_l (s_l = the forward's saved first-order sensitivities); MVSDF_DELTA_CHAIN=1 runs the 9-phase chain of GEMMs it replaced.  Same
    mathematics, different rounding: the parameter gradients of a whole step agree to ~1e-6 of their largest entry."""
    import tempfile
    import torch
    code = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
from test_gpu_native_step import _run
outs, losses, g, _, _ = _run(True, W=256, B=4, P=256, V=3, tp=0.3, sink=True)
torch.save(g.cpu(), sys.argv[1])
''' % (ROOT, ROOT)
    res = []
    with tempfile.TemporaryDirectory() as td:
        for chain in ('0', '1'):
            out = os.path.join(td, 'g%s.pt' % chain)
            e = dict(os.environ, MVSDF_DELTA_CHAIN=chain, MVSDF_LIB=dev_lib)
            p = subprocess.run([sys.executable, '-c', code, out], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            assert p.returncode == 0, p.stdout.decode(errors='replace')[-2000:]
            res.append(torch.load(out))
    a, b = res
    dev = float((a - b).abs().max() / b.abs().max())
    print('delta by scaling vs delta chain: max |dgrad| = %.3g of the largest gradient entry' % dev)
    assert 0 < dev < 2e-5 or dev == 0.0


def test_sample_rows_beside_the_tracer_change_no_bit():
    """The step may evaluate its E sample rows (eikonal / depth samples: idr.py:240-275) on a side stream while the tracer runs, and only the rays'
    rows after it (step_driver.hip, 1b).  Rows of the fused chain are independent of their tile: outputs, losses and gradients are identical bits."""
    import tempfile
    import torch
    code = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
from test_gpu_native_step import _run
res = []
for P in (256, 250):                                   # 250: E = 500 sample rows, not a multiple of the 16-row tile -- the rays' launch starts mid-tile
    outs, losses, g, _, _ = _run(True, W=256, B=4, P=P, V=3, tp=0.3, sink=True)
    res.append({'g': g.cpu(), 'losses': {k: v.detach().cpu() for k, v in losses.items()},
                'outs': {k: v.detach().cpu() for k, v in outs.items() if torch.is_tensor(v)}})
torch.save(res, sys.argv[1])
''' % (ROOT, ROOT)
    res = []
    with tempfile.TemporaryDirectory() as td:
        for split in ('0', '1'):
            out = os.path.join(td, 'r%s.pt' % split)
            e = dict(os.environ, MVSDF_SPLIT_ROWS=split)
            p = subprocess.run([sys.executable, '-c', code, out], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            assert p.returncode == 0, p.stdout.decode(errors='replace')[-2000:]
            res.append(torch.load(out))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a['g'], b['g'])
        for k in a['losses']:
            assert torch.equal(a['losses'][k], b['losses'][k]), k
        assert a['outs'].keys() == b['outs'].keys() and len(a['outs']) >= 8
        for k in a['outs']:
            assert torch.equal(a['outs'][k], b['outs'][k]), k
