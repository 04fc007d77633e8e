"""The native step driver (csrc/step_driver.hip: the training forward, the loss and the backward as one C call each) against the
Python-orchestrated route over the same kernels: same launches in the same order on the same inputs, so every output, every loss term and
every gradient entry must be IDENTICAL, bit for bit -- through the gradient sink and through plain autograd, in phase 0 (depth-surface
samples, geometry detached), with an object mask, with no hit at all, for an upstream on a term other than the total, and with two forwards
before the first backward.  (The reference-fixture tests in test_gpu_idr.py run through the native driver by default.)"""
import numpy as np
import pytest
import torch

from helpers import t
from mvsdf_amd import functional as Fn
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.model.loss import IDRLoss
from mvsdf_amd.optim import FlatAdam
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict

pytestmark = pytest.mark.gpu


def _model(W, native, skip_in=(4,)):
    m = IDRNetwork(ConfigDict(synth.model_conf(W, skip_in=skip_in)))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0, skip_in=skip_in).items()})
    m = m.cuda().train()
    m.native_step = native
    return m


def _batch(B, P, V, seed=3, focal_scale=1.4, phase0=False):
    inp, gt = synth.make_batch(B, P, V, seed=seed, feat_hw=(60, 80), focal_scale=focal_scale)
    if phase0:
        inp['depths'] = gt['depths'] = synth.make_depth_maps(inp['depth_cams'], 2.0, (0.0, 0.0, 0.0), seed=seed, hole_frac=0.05)
    return {k: t(v) for k, v in inp.items()}, {k: t(v) for k, v in gt.items()}


def _run(native, W=64, B=2, P=300, V=2, tp=0.3, sink=False, term='loss', focal_scale=1.4, skip_in=(4,), object_mask=None, phase0=False):
    m = _model(W, native, skip_in)
    inp, gt = _batch(B, P, V, focal_scale=focal_scale, phase0=phase0)
    if object_mask is not None:
        inp['object_mask'] = object_mask
    loss_fn = IDRLoss()
    loss_fn.native = native
    opt = FlatAdam(m.parameters(), lr=0.0) if sink else None
    torch.manual_seed(11)
    out = m(inp, tp)
    lo = loss_fn(out, dict(gt), tp, B)
    if sink:
        opt.zero_grad()
        opt.backward(lo[term])
    else:
        m.zero_grad()
        lo[term].backward()
    torch.cuda.synchronize()
    g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()]).clone()
    outs = {k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}
    return outs, {k: v.detach().clone() for k, v in lo.items()}, g, torch.rand(3), m


def _same(a, b, what):
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    assert torch.equal(a, b), '%s differs: max |d| = %g' % (what, float((a.float() - b.float()).abs().max()))


def _compare(**kw):
    o_n, l_n, g_n, r_n, m_n = _run(True, **kw)
    o_p, l_p, g_p, r_p, _ = _run(False, **kw)
    assert getattr(m_n, '_last_step', None) is not None, 'the native driver did not run'
    assert set(o_n) == set(o_p)
    for k in o_p:
        _same(o_n[k], o_p[k], k)
    for k in l_p:
        _same(l_n[k], l_p[k], k)
    _same(g_n, g_p, 'gradient')
    assert torch.equal(r_n, r_p)                                   # same consumption of torch's CPU generator
    assert float(g_p.abs().max()) > 0
    return o_n, l_n, g_n


@pytest.mark.parametrize('sink', [False, True])
@pytest.mark.parametrize('W,tp', [(64, 0.3), (64, 0.6), (256, 0.3)])
def test_native_equals_python_route(W, tp, sink):
    o, _, _ = _compare(W=W, tp=tp, sink=sink)
    n_hit = int(o['network_object_mask'].sum())
    assert 0 < n_hit < o['network_object_mask'].numel()


def test_native_phase0_with_depth_surface_samples():
    """train_progress < 1/6: depth-surface groups in both terms, geometry detached in front of the rendering net (idr.py:226-247,331-334)."""
    o, l, _ = _compare(W=64, tp=0.1, phase0=True, sink=True)
    R = o['network_object_mask'].numel()
    assert o['eikonal_output'].shape[1] == int(o['network_object_mask'].sum()) + R // 2 + 2 * (R // 2)
    assert float(l['feat_loss']) == 0.0 and float(l['surf_loss']) == 0.0


def test_native_several_skips_and_object_mask():
    om = (torch.rand(2, 300, generator=torch.Generator().manual_seed(5)) < 0.7).cuda()
    _compare(W=64, skip_in=(3, 6), object_mask=om, sink=False)


def test_native_upstream_on_another_term():
    """backward() from a term other than the total: the loss node's backward scales by (g_loss * weight + g_term)."""
    _compare(term='eikonal_loss')
    _compare(term='rgb_loss', sink=True)


def test_native_no_hit_at_all():
    """Every ray misses (cameras looking at the object through a tiny focal length miss it entirely when the object mask is empty and the
    rays are pushed off the sphere): N = 0 takes the route without the rendering-net backward."""
    m_n, m_p = _model(64, True), _model(64, False)
    res = []
    for m in (m_n, m_p):
        inp, gt = _batch(2, 128, 2)
        inp['uv'] = inp['uv'] * 0 + 5000.0                         # far outside the image: no ray meets the unit sphere
        loss_fn = IDRLoss()
        loss_fn.native = m.native_step
        torch.manual_seed(3)
        out = m(inp, 0.3)
        assert int(out['network_object_mask'].sum()) == 0 and out['diff_surf_pts'].shape == (0, 3)
        lo = loss_fn(out, dict(gt), 0.3, 2)
        m.zero_grad()
        lo['loss'].backward()
        res.append((lo, torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()])))
    for k in res[0][0]:
        _same(res[0][0][k].detach(), res[1][0][k].detach(), k)
    _same(res[0][1], res[1][1], 'gradient')


def test_two_forwards_before_the_first_backward():
    """Each forward owns its block: the backward of an earlier forward still sees its own saved activations after a later forward ran, and
    outputs kept from an earlier step do not change."""
    m = _model(64, True)
    inp_a, gt_a = _batch(2, 200, 2, seed=3)
    inp_b, gt_b = _batch(2, 200, 2, seed=4)
    loss_fn = IDRLoss()

    def grads(order):
        torch.manual_seed(5)
        out_a = m(inp_a, 0.3)
        pts_a = out_a['points'].clone()
        lo_a = loss_fn(out_a, dict(gt_a), 0.3, 2)
        if order == 'interleaved':
            out_b = m(inp_b, 0.3)                                  # a second forward before a's backward
            assert torch.equal(out_a['points'], pts_a)
            lo_b = loss_fn(out_b, dict(gt_b), 0.3, 2)
        m.zero_grad()
        lo_a['loss'].backward()
        return torch.cat([p.grad.flatten() for p in m.parameters()]).clone()

    _same(grads('interleaved'), grads('alone'), 'gradient of the first forward')


@pytest.mark.parametrize('direct', [True, False])
def test_a_backward_after_an_optimizer_step_in_between_raises(direct):
    """forward A, forward B, backward B, FlatAdam.step() (which writes the parameters through a raw pointer), backward A: A's saved activations belong to the
    OLD weights while its weight-norm fold backward would read the NEW ones -- the step must raise like autograd does after torch.optim.Adam.step(), through
    FlatAdam.backward's direct route and through loss.backward() alike (FlatAdam.step bumps the parameters' version counters)."""
    m = _model(64, True)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    inp_a, gt_a = _batch(2, 200, 2, seed=3)
    inp_b, gt_b = _batch(2, 200, 2, seed=4)
    loss_fn = IDRLoss()
    torch.manual_seed(5)
    lo_a = loss_fn(m(inp_a, 0.3), dict(gt_a), 0.3, 2)
    lo_b = loss_fn(m(inp_b, 0.3), dict(gt_b), 0.3, 2)
    v0 = [p._version for p in m.parameters()]
    opt.zero_grad()
    opt.backward(lo_b['loss']) if direct else lo_b['loss'].backward()
    opt.step()
    assert all(p._version > v for p, v in zip(m.parameters(), v0))
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        opt.backward(lo_a['loss']) if direct else lo_a['loss'].backward()
    # and the ordinary order keeps working after it
    opt.zero_grad()
    lo_c = loss_fn(m(inp_a, 0.3), dict(gt_a), 0.3, 2)
    opt.backward(lo_c['loss'])
    opt.step()
    assert torch.isfinite(torch.cat([p.flatten() for p in m.parameters()])).all()


def test_native_loss_accepts_foreign_outputs():
    """IDRLoss's fused route is generic: it takes any output dict (here: tensors made by plain torch ops with autograd leaves)."""
    m = _model(64, True)
    inp, gt = _batch(2, 200, 2)
    torch.manual_seed(1)
    out = m(inp, 0.3)
    leaf = {k: out[k].detach().clone().requires_grad_(True) for k in ('diff_surf_pts', 'rgb_values', 'grad_theta', 'eikonal_output', 'surf_indicator_output')}
    o2 = dict(out)
    o2.update(leaf)
    o2['eikonal_points_hom'] = out['eikonal_points_hom'].clone()
    o3 = dict(o2)
    o3['eikonal_points_hom'] = out['eikonal_points_hom'].clone()
    a, b = IDRLoss(), IDRLoss()
    a.native, b.native = True, False
    la = a(o2, dict(gt), 0.3, 2)
    ga = torch.autograd.grad(la['loss'], list(leaf.values()))
    lb = b(o3, dict(gt), 0.3, 2)
    gb = torch.autograd.grad(lb['loss'], list(leaf.values()))
    for k in la:
        _same(la[k].detach(), lb[k].detach(), k)
    for x, y, k in zip(ga, gb, leaf):
        _same(x, y, 'd loss / d ' + k)
    _same(o2['eikonal_points_hom'], o3['eikonal_points_hom'], 'world-space side effect on eikonal_points_hom')


def test_host_time_of_a_native_step():
    """The point of the driver: the interpreter's share of a step.  Measured here with a tiny batch (GPU work ~ nothing) so that the wall
    time per step IS the host time; printed, and bounded loosely (the box's CPU is shared)."""
    import time
    res = {}
    for native in (True, False):
        m = _model(64, native)
        inp, gt = _batch(2, 64, 2)
        loss_fn = IDRLoss()
        loss_fn.native = native
        opt = FlatAdam(m.parameters(), lr=0.0)

        def step():
            opt.zero_grad()
            out = m(inp, 0.3)
            lo = loss_fn(out, dict(gt), 0.3, 2)
            opt.backward(lo['loss'])
            opt.step(grad_cap=2.0)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        res[native] = (time.perf_counter() - t0) / 100 * 1e3
    print('wall ms per tiny step: native %.3f, python route %.3f' % (res[True], res[False]))
    assert res[True] < res[False]
