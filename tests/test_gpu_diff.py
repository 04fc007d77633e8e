"""GPU parity of the differentiable passes (value + normal, double backward, rendering net) vs goldens and the numpy oracle."""
import numpy as np
import pytest
import torch

from conftest import golden
from helpers import sdf_packed_net, t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
from oracle import oracle_np as ON

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['x3', 'f32'])
def chain_arithmetic(request, monkeypatch):
    """Every test of this file runs on both forms of the fused SDF chains: 'x3' (the product: three bf16 terms per fp32 value on v_mfma_f32_16x16x32_bf16,
    csrc/chain_x3.h -- taken when the network carries the wx3 packs) and 'f32' (the fp32-input MFMA chains: networks packed without them)."""
    monkeypatch.setattr(ops, 'CHAIN_X3', request.param == 'x3')
    return request.param


def _rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def render_packed_net(sd):
    return sdf_packed_net(sd, prefix='rendering_network', skip_layer=-1, multires=0)


@pytest.mark.parametrize('W', [64, 256, 512])
def test_sdf_value_normal_vs_golden(W):
    g = golden('sdf_w%d' % W)
    net = sdf_packed_net(synth.make_state_dict(W, int(g['seed'])))
    x = t(g['x'])
    y, n, _ = ops.sdf_forward(net, x, x.shape[0])
    np.testing.assert_allclose(y.cpu().numpy(), g['out'], rtol=1e-4, atol=3e-6)
    assert _rel(n, g['grad']) < 5e-5
    y2, n2, _ = ops.sdf_forward(net, x, 37)                    # normals on a prefix only (ragged last tile)
    assert torch.equal(y2, y) and torch.equal(n2, n[:37])
    y3, _, _ = ops.sdf_forward(net, x[:5], 0)
    assert torch.equal(y3, y[:5])


def test_sdf_double_backward_vs_golden():
    g = golden('sdf_bwd_w64')
    sd = synth.make_state_dict(64, int(g['seed']))
    net = sdf_packed_net(sd)
    x = t(g['x'])
    M = x.shape[0]
    y, n, ctx = ops.sdf_forward(net, x, M)
    dWs, dbs, dx = ops.sdf_backward(net, x, M, M, M, t(g['dy']), t(g['dn']), ctx, True)
    assert _rel(dx, g['dx']) < 2e-4
    for l, (dW, db) in enumerate(zip(dWs, dbs)):
        v = t(sd['implicit_network.lin%d.weight_v' % l]); gg = t(sd['implicit_network.lin%d.weight_g' % l])
        dv, dg = ops.fold_backward(v, gg, dW.contiguous())
        assert _rel(dv, g['d_lin%d.weight_v' % l]) < 5e-4, l
        assert _rel(dg, g['d_lin%d.weight_g' % l]) < 5e-4, l
        assert _rel(db, g['d_lin%d.bias' % l]) < 5e-4, l
    # value-only backward (dn = None) and a row prefix
    _, _, dx1 = ops.sdf_backward(net, x, M, M, M, t(g['dy']), None, ctx, True)
    assert _rel(dx1, g['dx_value_only']) < 2e-4
    Mb = 70
    onet = ON.sdf_net(sd)
    _, _, cache = ON.sdf_forward(onet, g['x'][:Mb])
    oW, ob, odx = ON.sdf_backward(onet, cache, g['dy'][:Mb], g['dn'][:Mb])
    dWs2, dbs2, dx2 = ops.sdf_backward(net, x, M, M, Mb, t(g['dy'][:Mb]), t(g['dn'][:Mb]), ctx, True)
    assert _rel(dx2, odx) < 2e-4
    for l in range(len(dWs2)):
        assert _rel(dWs2[l], oW[l]) < 5e-4 and _rel(dbs2[l], ob[l]) < 5e-4


def test_sdf_backward_w256_vs_oracle():
    sd = synth.make_state_dict(256, 0)
    net, onet = sdf_packed_net(sd), ON.sdf_net(sd)
    rs = np.random.RandomState(4)
    M, Mg = 1100, 700
    x = rs.uniform(-1, 1, size=(M, 3)).astype(np.float32)
    dy = (rs.normal(size=(M, 258)) * 0.1).astype(np.float32)
    dn = rs.normal(size=(Mg, 3)).astype(np.float32)
    y, n, ctx = ops.sdf_forward(net, t(x), Mg)
    oy, on, cache = ON.sdf_forward(onet, x)
    assert _rel(y, oy) < 2e-5 and _rel(n, on[:Mg]) < 5e-5
    # rows [0, Mg) carry dn; evaluate the oracle in two parts (linearity)
    dn_full = np.zeros((M, 3)); dn_full[:Mg] = dn
    oW, ob, odx = ON.sdf_backward(onet, cache, dy, dn_full)
    dWa, dba, dxa = ops.sdf_backward(net, t(x), M, Mg, Mg, t(dy[:Mg]), t(dn), ctx, True)
    dy_rest = dy.copy(); dy_rest[:Mg] = 0
    dWb, dbb, dxb = ops.sdf_backward(net, t(x), M, Mg, M, t(dy_rest), None, ctx, True)
    for l in range(9):
        assert _rel(dWa[l] + dWb[l], oW[l]) < 5e-4, l
        assert _rel(dba[l] + dbb[l], ob[l]) < 5e-4, l
    dx = dxb.clone(); dx[:Mg] += dxa
    assert _rel(dx, odx) < 5e-4


def test_render_forward_backward_vs_golden():
    g = golden('render_bwd_w64')
    sd = synth.make_state_dict(64, int(g['seed']))
    net = render_packed_net(sd)
    N = g['points'].shape[0]
    wide = torch.zeros(N, 258, device='cuda'); wide[:, 2:] = t(g['feat'])         # features as a column slice (ld = 258)
    rgb, ctx = ops.render_forward(net, t(g['points']), t(g['view']), t(g['normals']), wide[:, 2:], 4)
    np.testing.assert_allclose(rgb.cpu().numpy(), g['rgb'], rtol=1e-4, atol=2e-6)
    dWs, dbs, din = ops.render_backward(net, N, t(g['drgb']), ctx)
    dv = 3 + 6 * 4
    assert _rel(din[:, :3], g['dpoints']) < 2e-4 and _rel(din[:, 3 + dv:6 + dv], g['dnormals']) < 2e-4
    assert _rel(din[:, 6 + dv:], g['dfeat']) < 2e-4
    for l, (dW, db) in enumerate(zip(dWs, dbs)):
        v = t(sd['rendering_network.lin%d.weight_v' % l]); gg = t(sd['rendering_network.lin%d.weight_g' % l])
        dvv, dg = ops.fold_backward(v, gg, dW.contiguous())
        assert _rel(dvv, g['d_lin%d.weight_v' % l]) < 5e-4 and _rel(dg, g['d_lin%d.weight_g' % l]) < 5e-4
        assert _rel(db, g['d_lin%d.bias' % l]) < 5e-4


@pytest.mark.parametrize('W', [256, 512])
def test_two_tile_chain_workgroups_equal_one_tile(W):
    """257..512 row tiles per launch run the fused chain kernels with two 16-row tiles per workgroup (diff_mlp.hip, mv_chain_mt / mv_chain_mt_x3; hidden
    width 512: the x3 chains' 8-wave x 4-column-tile form): rows are independent, so forward outputs, saved context use and input adjoints must equal, bit
    for bit, those of launches small enough to use one tile per workgroup; the weight gradients agree to summation order."""
    sd = synth.make_state_dict(W, 0)
    net = sdf_packed_net(sd)
    M = 4500                                                     # 282 tiles -> two per workgroup;  1500 rows (94 tiles) -> one
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(M, 3, generator=gen) * 2 - 1).cuda()
    dy = torch.randn(M, net.layers[-1].N, generator=gen).cuda() * 0.1
    dn = torch.randn(M, 3, generator=gen).cuda()
    y, n, ctx = ops.sdf_forward(net, x, M)
    dWs, dbs, dx = ops.sdf_backward(net, x, M, M, M, dy, dn, ctx, True)
    wsA, dxX = ops.sdf_backward_pair(net, M, M, M, dy, dn, 100, 2000, dy[100:2100].contiguous(), dn[100:2100].contiguous(), ctx)
    fbar = torch.randn(2000, generator=gen).cuda()
    dy2 = dy.clone(); dy2[100:2100, 0] += fbar
    dWf, dbf = ops.sdf_backward_finish(net, M, M, M, dy2, ctx, wsA, 100, 2000, fbar)
    acc_W = [torch.zeros_like(w) for w in dWs]
    acc_Wf = [torch.zeros_like(w) for w in dWs]
    for r0 in range(0, M, 1500):
        xs = x[r0:r0 + 1500].contiguous()
        ys, ns, cs = ops.sdf_forward(net, xs, 1500)
        assert torch.equal(ys, y[r0:r0 + 1500]) and torch.equal(ns, n[r0:r0 + 1500])
        dWp, _, dxp = ops.sdf_backward(net, xs, 1500, 1500, 1500, dy[r0:r0 + 1500].contiguous(), dn[r0:r0 + 1500].contiguous(), cs, True)
        assert torch.equal(dxp, dx[r0:r0 + 1500])
        dWq, _, _ = ops.sdf_backward(net, xs, 1500, 1500, 1500, dy2[r0:r0 + 1500].contiguous(), dn[r0:r0 + 1500].contiguous(), cs, False)
        for l in range(len(dWs)):
            acc_W[l] += dWp[l]; acc_Wf[l] += dWq[l]
    assert torch.equal(dxX, dx[100:2100])                        # pass X of the pair = the same input adjoint
    for l in range(len(dWs)):
        assert _rel(acc_W[l], dWs[l].cpu().numpy()) < 2e-5, l
        assert _rel(acc_Wf[l], dWf[l].cpu().numpy()) < 2e-5, l   # pair + delta pass + finish = plain backward with the fbar folded into dy


def test_backward_rejects_upstream_of_the_wrong_width():
    net = sdf_packed_net(synth.make_state_dict(64, 0))
    x = torch.rand(40, 3).cuda()
    y, n, ctx = ops.sdf_forward(net, x, 40)
    with pytest.raises(ValueError, match='dy must be'):
        ops.sdf_backward(net, x, 40, 40, 40, torch.zeros(40, y.shape[1] - 1).cuda(), None, ctx, True)
    with pytest.raises(ValueError, match='dn must be'):
        ops.sdf_backward(net, x, 40, 40, 40, torch.zeros_like(y), torch.zeros(39, 3).cuda(), ctx, True)


def test_transposed_three_term_pack_equals_the_pack_of_the_transposed_matrix():
    """mvsdf_pack_bf16x3t_net (the W_l^T packs of the x3 chains, MvsdfNetDesc.wx3 of the transposed descriptor) must write what mvsdf_pack_bf16x3_net writes
    for the explicitly transposed matrix -- ragged shapes included (39 x 64, 217 x 256, 258 x 256: partial column tiles and k-blocks, zero padding)."""
    import ctypes as C
    from mvsdf_amd._lib import check, lib, stream_of
    gen = torch.Generator().manual_seed(3)
    ws = [torch.randn(n, k, generator=gen).cuda() for n, k in ((64, 39), (217, 256), (258, 256), (3, 64))]
    wts = [w.t().contiguous() for w in ws]
    n = len(ws)
    N = (C.c_int * n)(*[w.shape[0] for w in ws]); K = (C.c_int * n)(*[w.shape[1] for w in ws])
    Nt = (C.c_int * n)(*[w.shape[0] for w in wts]); Kt = (C.c_int * n)(*[w.shape[1] for w in wts])
    size = lambda nn, kk: 3 * lib().mvsdf_packed_bf16_bytes(nn, kk, 0)
    a = [torch.full((size(w.shape[1], w.shape[0]),), 0x5a, dtype=torch.uint8, device='cuda') for w in ws]       # pack of W^T through the transposing entry point
    b = [torch.full((size(w.shape[0], w.shape[1]),), 0xa5, dtype=torch.uint8, device='cuda') for w in wts]     # pack of the explicit transpose
    s = stream_of(ws[0])
    check(lib().mvsdf_pack_bf16x3t_net(n, ops._ptr_array(ws), N, K, ops._ptr_array(a), s), 'mvsdf_pack_bf16x3t_net')
    check(lib().mvsdf_pack_bf16x3_net(n, ops._ptr_array(wts), Nt, Kt, ops._ptr_array(b), s), 'mvsdf_pack_bf16x3_net')
    for l in range(n):
        assert a[l].numel() == b[l].numel() and torch.equal(a[l], b[l]), l
