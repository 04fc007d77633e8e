"""Constructor options of the reference that the shipped conf does not use (idr.py:20-31, 110-119): weight_norm=False, rendering modes
'no_view_dir' / 'no_normal', multires_view=0, multires=0 -- each against a plain PyTorch fp32 restatement of the same network with autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import t
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork, ImplicitNetwork, RenderingNetwork
from mvsdf_amd.model.loss import IDRLoss
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict

pytestmark = pytest.mark.gpu


def _pe(x, multires):
    out = [x]
    for m in range(multires):
        out += [torch.sin(x * 2.0 ** m), torch.cos(x * 2.0 ** m)]
    return torch.cat(out, -1)


def _torch_render(net, points, normals, view, feat):
    v = _pe(view, net.multires_view) if net.multires_view > 0 else view
    x = {'idr': [points, v, normals, feat], 'no_view_dir': [points, normals, feat], 'no_normal': [points, v, feat]}[net.mode]
    x = torch.cat(x, -1)
    n = net.num_layers - 1
    for l in range(n):
        lin = getattr(net, 'lin%d' % l)
        x = F.linear(x, lin.weight, lin.bias)
        if l < n - 1:
            x = torch.relu(x)
    return torch.tanh(x)


@pytest.mark.parametrize('mode,d_in,mv,wn', [('idr', 9, 4, True), ('idr', 9, 0, True), ('no_view_dir', 6, 0, True), ('no_normal', 6, 4, True),
                                            ('no_normal', 6, 0, False), ('idr', 9, 4, False)])
def test_rendering_network_modes_vs_torch(mode, d_in, mv, wn):
    torch.manual_seed(1)
    net = RenderingNetwork(256, mode, d_in, 3, [64, 64], weight_norm=wn, multires_view=mv).cuda()
    keys = list(net.state_dict().keys())
    assert keys[:2] == (['lin0.bias', 'lin0.weight_g'] if wn else ['lin0.weight', 'lin0.bias'])     # nn.Linear / weight_norm key layout
    N = 300
    g = torch.Generator().manual_seed(2)
    mk = lambda *s: torch.randn(*s, generator=g).cuda().requires_grad_(True)
    pts, nrm, feat = mk(N, 3), mk(N, 3), mk(N, 256)
    view = F.normalize(torch.randn(N, 3, generator=g), dim=1).cuda()
    drgb = torch.randn(N, 3, generator=g).cuda()
    rgb = net(pts, nrm, view, feat)
    params = list(net.parameters())
    got = torch.autograd.grad((rgb * drgb).sum(), [pts, nrm, feat] + params, allow_unused=True)
    ref_rgb = _torch_render(net, pts, nrm, view, feat)
    ref = torch.autograd.grad((ref_rgb * drgb).sum(), [pts, nrm, feat] + params, allow_unused=True)
    assert torch.allclose(rgb, ref_rgb, atol=2e-6, rtol=1e-5)
    for i, (a, b) in enumerate(zip(got, ref)):
        if b is None:                                                   # an input the mode does not use
            assert a is None or float(a.abs().max()) == 0.0
            continue
        assert a is not None and torch.allclose(a, b, rtol=2e-4, atol=2e-6 * float(b.abs().max()) + 1e-9), i
    with pytest.raises(ValueError):
        RenderingNetwork(256, 'no_view_dir', 9, 3, [64], multires_view=0)      # d_in does not match the mode's inputs


def _torch_sdf(net, x):
    inp = _pe(x, net.multires) if net.multires > 0 else x
    h = inp
    n = net.num_layers - 1
    for l in range(n):
        lin = getattr(net, 'lin%d' % l)
        if l in net.skip_in:
            h = torch.cat([h, inp], 1) / np.sqrt(2)
        h = F.linear(h, lin.weight, lin.bias)
        if l < n - 1:
            h = F.softplus(h, beta=100)
    return h


@pytest.mark.parametrize('multires,wn', [(0, True), (6, False), (0, False)])
def test_implicit_network_options_vs_torch(multires, wn):
    torch.manual_seed(3)
    net = ImplicitNetwork(256, 3, 1, [64] * 8, geometric_init=True, bias=0.6, skip_in=[4], weight_norm=wn, multires=multires).cuda()
    with torch.no_grad():                                                # away from the bare geometric init: every weight matters
        for p in net.parameters():
            p.add_(0.02 * p.abs().mean() * torch.randn_like(p))
    x = (torch.rand(500, 3, generator=torch.Generator().manual_seed(4)) * 2 - 1).cuda()
    xg = x.clone()
    y = net(x)
    n = net.gradient(xg)[:, 0]
    xr = x.clone().requires_grad_(True)
    yr = _torch_sdf(net, xr)
    nr = torch.autograd.grad(yr[:, 0].sum(), xr, create_graph=True)[0]
    assert torch.allclose(y, yr, rtol=1e-4, atol=5e-6) and torch.allclose(n, nr, rtol=2e-4, atol=2e-5)
    # first- and second-order parameter gradients
    dy, dn = torch.randn_like(y) * 0.1, torch.randn_like(n)
    params = list(net.parameters())
    x2 = x.clone()
    y2 = net(x2)
    n2 = net.gradient(x2)[:, 0]
    got = torch.autograd.grad((y2 * dy).sum() + (n2 * dn).sum(), params)
    ref = torch.autograd.grad((yr * dy).sum() + (nr * dn).sum(), params)
    for (k, _), a, b in zip(net.named_parameters(), got, ref):
        assert torch.allclose(a, b, rtol=2e-3, atol=1e-3 * float(b.abs().max()) + 1e-8), k
    # the tracer's column-0 kernel agrees with the differentiable forward
    from mvsdf_amd import ops
    y0 = ops.sdf_col0(net.native_sdf().native_net, x)
    assert torch.allclose(y0, y[:, 0], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize('wide', [288, 304])
def test_skip_layer_wider_than_the_hidden_layers_vs_torch(wide):
    """dims[skip] above the other hidden widths (hidden 256, dims[4] = 288 / 304: the skip layer takes 288 / 304 inputs = 18 / 19 column tiles of its W^T phases,
    more than the 16 the one-tile-per-wave chain forms cover): the fused chains must pick a form wide enough for EVERY phase (capi_util.h::mv_chain_ntw looks
    at the layers' inputs too) -- value, normal, and first / second-order parameter gradients against the plain PyTorch restatement."""
    torch.manual_seed(3)
    dims = [256, 256, 256, wide, 256, 256, 256, 256]
    net = ImplicitNetwork(256, 3, 1, dims, geometric_init=True, bias=0.6, skip_in=[4], weight_norm=True, multires=6).cuda()
    assert net.lin3.weight_v.shape[0] == wide - 39 and net.lin4.weight_v.shape[1] == wide
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.02 * p.abs().mean() * torch.randn_like(p))
    x = (torch.rand(700, 3, generator=torch.Generator().manual_seed(4)) * 2 - 1).cuda()
    y = net(x)
    n = net.gradient(x.clone())[:, 0]
    xr = x.clone().requires_grad_(True)
    yr = _torch_sdf(net, xr)
    nr = torch.autograd.grad(yr[:, 0].sum(), xr, create_graph=True)[0]
    assert torch.allclose(y, yr, rtol=1e-4, atol=5e-6) and torch.allclose(n, nr, rtol=2e-4, atol=2e-5)
    dy, dn = torch.randn_like(y) * 0.1, torch.randn_like(n)
    params = list(net.parameters())
    x2 = x.clone()
    got = torch.autograd.grad((net(x2) * dy).sum() + (net.gradient(x2)[:, 0] * dn).sum(), params)
    ref = torch.autograd.grad((yr * dy).sum() + (nr * dn).sum(), params)
    for (k, _), a, b in zip(net.named_parameters(), got, ref):
        assert torch.allclose(a, b, rtol=2e-3, atol=1e-3 * float(b.abs().max()) + 1e-8), k
    from mvsdf_amd import ops
    y0 = ops.sdf_col0(net.native_sdf().native_net, x)
    assert torch.allclose(y0, y[:, 0], rtol=1e-5, atol=2e-6)


def test_idr_step_without_weight_norm_and_without_normals():
    """A whole training step with weight_norm=False in both networks and rendering mode 'no_normal': same outputs as the weight-normed
    model carrying the same folded weights; gradients chain through the fold (dW of the plain model -> dv, dg of the normed one)."""
    from mvsdf_amd import ops
    W = 64
    conf = synth.model_conf(W)
    conf['rendering_network'].update(mode='no_normal', d_in=6)
    ref = IDRNetwork(ConfigDict(conf)).cuda().train()
    torch.manual_seed(5)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0).items() if k.startswith('implicit')}
    ref.implicit_network.load_state_dict({k[len('implicit_network.'):]: v for k, v in sd.items()})
    conf2 = synth.model_conf(W)
    conf2['rendering_network'].update(mode='no_normal', d_in=6, weight_norm=False)
    conf2['implicit_network'].update(weight_norm=False)
    plain = IDRNetwork(ConfigDict(conf2)).cuda().train()
    with torch.no_grad():
        for netname in ('implicit_network', 'rendering_network'):
            a, b = getattr(ref, netname), getattr(plain, netname)
            for l in range(a.num_layers - 1):
                la, lb = getattr(a, 'lin%d' % l), getattr(b, 'lin%d' % l)
                w, _, _ = ops.fold_pack(la.weight_v.detach(), la.weight_g.detach())   # the exact folded weights the kernels use
                lb.weight.copy_(w); lb.bias.copy_(la.bias)
    inp, gt = synth.make_batch(2, 128, 2, seed=3, feat_hw=(60, 80), focal_scale=1.4)
    inp, gt = {k: t(v) for k, v in inp.items()}, {k: t(v) for k, v in gt.items()}
    outs = {}
    for name, m in (('ref', ref), ('plain', plain)):
        torch.manual_seed(7)
        out = m(inp, 0.3)
        lo = IDRLoss()(out, dict(gt), 0.3, 2)
        lo['loss'].backward()
        outs[name] = (out, lo)
    for k in ('rgb_values', 'grad_theta', 'diff_surf_pts', 'network_object_mask'):
        assert torch.equal(outs['ref'][0][k], outs['plain'][0][k]), k
    assert float(outs['ref'][1]['loss']) == float(outs['plain'][1]['loss'])
    for netname in ('implicit_network', 'rendering_network'):
        a, b = getattr(ref, netname), getattr(plain, netname)
        for l in range(a.num_layers - 1):
            la, lb = getattr(a, 'lin%d' % l), getattr(b, 'lin%d' % l)
            dv, dg = ops.fold_backward(la.weight_v.detach(), la.weight_g.detach(), lb.weight.grad)
            assert torch.allclose(dv, la.weight_v.grad, rtol=1e-5, atol=1e-8) and torch.allclose(dg, la.weight_g.grad, rtol=1e-5, atol=1e-8)
            assert torch.allclose(lb.bias.grad, la.bias.grad, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('name', ['sdf_bwd_w64_skips36', 'sdf_bwd_w64_skip8', 'sdf_bwd_w64_skips48'])
def test_several_skip_connections(name):
    """skip_in = (3, 6), (8,) = a skip into the LAST Linear (idr.py:46-49,86) and (4, 8): (1) value / normal / double backward of the HIP chains
    vs the reference golden, (2) the tracer bit for bit vs the C oracle in fp32 on every engine, (3) the bf16-weight split-activation engine against
    the oracle on rounded weights, (4) the f32x3 engine bit for bit vs its instruction-model oracle, (5) the layer shapes of the module."""
    from conftest import golden
    from helpers import sdf_packed_net, trace_params
    from mvsdf_amd import ops
    from oracle import oracle as O
    g = golden(name)
    skips = tuple(int(v) for v in g['skip_in'])
    sd = synth.make_state_dict(64, int(g['seed']), skip_in=skips)
    net = sdf_packed_net(sd, skip_layer=skips)
    rel = lambda a, b: float(np.abs(a.detach().cpu().numpy() - b).max() / max(np.abs(b).max(), 1e-12))
    x = t(g['x'])
    M = x.shape[0]
    y, n, ctx = ops.sdf_forward(net, x, M)
    np.testing.assert_allclose(y.cpu().numpy(), g['out'], rtol=1e-4, atol=3e-6)
    assert rel(n, g['grad']) < 5e-5
    dWs, dbs, dx = ops.sdf_backward(net, x, M, M, M, t(g['dy']), t(g['dn']), ctx, True)
    assert rel(dx, g['dx']) < 2e-4
    for l, (dW, db) in enumerate(zip(dWs, dbs)):
        v = t(sd['implicit_network.lin%d.weight_v' % l]); gg = t(sd['implicit_network.lin%d.weight_g' % l])
        dv, dg = ops.fold_backward(v, gg, dW.contiguous())
        assert rel(dv, g['d_lin%d.weight_v' % l]) < 5e-4, l
        assert rel(dg, g['d_lin%d.weight_g' % l]) < 5e-4, l
        assert rel(db, g['d_lin%d.bias' % l]) < 5e-4, l
    _, _, dx1 = ops.sdf_backward(net, x, M, M, M, t(g['dy']), None, ctx, True)
    assert rel(dx1, g['dx_value_only']) < 2e-4
    # (2) tracer vs the C oracle
    onet = O.Net(sd, skip_in=skips)
    inp, _ = synth.make_batch(2, 160, 0, seed=1, with_features=False, focal_scale=1.4)
    dirs, cam = ops.camera_rays(t(inp['uv']), t(inp['pose']), t(inp['intrinsics']))
    iv = torch.linspace(0, 1, 100)
    steps = np.random.RandomState(0).uniform(size=100).astype(np.float32)
    om = torch.ones(320, dtype=torch.bool, device='cuda')
    p_o, m_o, d_o, rows = O.trace(onet, cam.cpu().numpy(), dirs.cpu().numpy(), np.ones(320, bool), True, steps, iv.numpy(), **synth.model_conf(64)['ray_tracer'])
    for mt in (1, 2):
        pts, mask, dists, cnt = ops.trace(net, cam, dirs, om, trace_params(64), True, iv.cuda(), t(steps), mt=mt)
        assert np.array_equal(mask.cpu().numpy(), m_o) and np.array_equal(dists.cpu().numpy(), d_o) and np.array_equal(pts.cpu().numpy(), p_o), mt
        assert np.array_equal(cnt.cpu().numpy()[:4], rows)
    xs = (torch.rand(3000, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1).cuda()
    want = O.sdf_forward(onet, xs.cpu().numpy(), ncols=1)[:, 0]
    for mt in (1, 2, 4, 49):                                     # 1 / 2 / 4 row tiles per workgroup, the sphere tracer's carried weight ring
        assert np.array_equal(ops.sdf_col0(net, xs, mt=mt).cpu().numpy(), want), mt
    # (3) the bf16-weight engine with split activations vs the oracle on rounded weights (masks / values: tests/test_gpu_bf16s.py; here: the skip layout)
    ops.pack_bf16_net(net, terms=3)
    twin = O.sdf_forward(O.Net(sd, skip_in=skips, bf16='weights'), xs.cpu().numpy(), ncols=1)[:, 0]
    got = ops.sdf_col0(net, xs).cpu().numpy()
    assert np.abs(got - twin).max() < 4e-6
    assert np.abs(got - want).max() < 2e-2                       # and near the fp32 network
    # (4) the product default: fp32 weights and activations as three bf16 terms each, against the oracle's model of the matrix instruction
    net3 = sdf_packed_net(sd, skip_layer=skips)
    ops.pack_trace_net(net3, 'f32x3')
    o3 = O.Net(sd, skip_in=skips, bf16='f32x3')
    assert np.array_equal(ops.sdf_col0(net3, xs).cpu().numpy(), O.sdf_forward(o3, xs.cpu().numpy(), ncols=1)[:, 0])
    p_o, m_o, d_o, rows = O.trace(o3, cam.cpu().numpy(), dirs.cpu().numpy(), np.ones(320, bool), True, steps, iv.numpy(), **synth.model_conf(64)['ray_tracer'])
    pts, mask, dists, cnt = ops.trace(net3, cam, dirs, om, trace_params(64), True, iv.cuda(), t(steps))
    assert np.array_equal(mask.cpu().numpy(), m_o) and np.array_equal(dists.cpu().numpy(), d_o) and np.array_equal(pts.cpu().numpy(), p_o)
    assert np.array_equal(cnt.cpu().numpy()[:4], rows)
    # (5)
    m2 = ImplicitNetwork(256, 3, 1, [64] * 8, skip_in=skips, multires=6)
    for sk in skips:
        assert getattr(m2, 'lin%d' % (sk - 1)).weight_v.shape == (25, 64) and getattr(m2, 'lin%d' % sk).weight_v.shape[1] == 64
    assert m2.fold_spec()[3] == (skips if len(skips) > 1 else skips[0])
