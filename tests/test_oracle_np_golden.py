"""numpy oracle of the differentiable half (oracle/oracle_np.py) pinned against the PyTorch-reference goldens."""
import numpy as np
import pytest

from conftest import golden
from mvsdf_amd.utils import synth
from oracle import oracle_np as ON


def _sd(g):
    sd = synth.make_state_dict(int(g['W']), int(g['seed']))
    np.testing.assert_allclose(synth.state_checksum(sd), g['checksum'], rtol=0, atol=0)
    return sd


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


@pytest.mark.parametrize('W', [64, 256, 512])
def test_sdf_value_and_normal(W):
    g = golden('sdf_w%d' % W)
    net = ON.sdf_net(_sd(g))
    y, n, _ = ON.sdf_forward(net, g['x'])
    np.testing.assert_allclose(y, g['out'], rtol=1e-4, atol=3e-6)
    assert _rel(n, g['grad']) < 2e-5                      # ImplicitNetwork.gradient (idr.py:96-107)


def test_sdf_double_backward():
    g = golden('sdf_bwd_w64')
    sd = _sd(g)
    net = ON.sdf_net(sd)
    _, _, cache = ON.sdf_forward(net, g['x'])
    dW, db, dx = ON.sdf_backward(net, cache, g['dy'], g['dn'])
    assert _rel(dx, g['dx']) < 1e-4
    for l in range(net.n_layers):
        dv, dg = ON.fold_backward(net.v[l], net.g[l], dW[l])
        assert _rel(dv, g['d_lin%d.weight_v' % l]) < 2e-4, l
        assert _rel(dg, g['d_lin%d.weight_g' % l]) < 2e-4, l
        assert _rel(db[l], g['d_lin%d.bias' % l]) < 2e-4, l
    _, _, dx1 = ON.sdf_backward(net, cache, g['dy'], None)
    assert _rel(dx1, g['dx_value_only']) < 1e-4


def test_render_forward_backward():
    g = golden('render_bwd_w64')
    net = ON.render_net(_sd(g))
    rgb, cache = ON.render_forward(net, g['points'], g['normals'], g['view'], g['feat'])
    np.testing.assert_allclose(rgb, g['rgb'], rtol=1e-4, atol=2e-6)
    dW, db, dp, dn, df = ON.render_backward(net, cache, g['drgb'])
    assert _rel(dp, g['dpoints']) < 1e-4 and _rel(dn, g['dnormals']) < 1e-4 and _rel(df, g['dfeat']) < 1e-4
    for l in range(net.n_layers):
        dv, dg = ON.fold_backward(net.v[l], net.g[l], dW[l])
        assert _rel(dv, g['d_lin%d.weight_v' % l]) < 2e-4 and _rel(dg, g['d_lin%d.weight_g' % l]) < 2e-4
        assert _rel(db[l], g['d_lin%d.bias' % l]) < 2e-4
    g2 = golden('render_w64')
    rgb2, _ = ON.render_forward(ON.render_net(_sd(g2)), g2['points'], g2['normals'], g2['view'], g2['feat'])
    np.testing.assert_allclose(rgb2, g2['rgb'], rtol=1e-4, atol=2e-6)


def test_sample_network():
    g = golden('sample_network')
    out = ON.sample_network(*[g[k].astype(np.float64) for k in ('surface_output', 'surface_sdf_values', 'surface_points_grad',
                                                                'surface_dists', 'surface_cam_loc', 'surface_ray_dirs')])
    np.testing.assert_allclose(out, g['out'], rtol=2e-5, atol=1e-5)


def test_feat_corr_loss_and_gradient():
    g = golden('feat_corr')
    B, P, V = int(g['B']), int(g['P']), int(g['V'])
    _, gt = synth.make_batch(B, P, V, seed=int(g['seed']), size=float(g['scene_size']), center=tuple(g['scene_center']),
                             feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    counts = g['hits'].reshape(B, P).sum(1)
    loss = ON.feat_corr_loss(g['points'], counts, gt['feat'], gt['cam'], gt['feat_src'], gt['src_cams'], gt['size'][0], gt['center'][0])
    assert abs(loss - float(g['loss'])) < 2e-6
    sub = slice(0, 12)                                       # finite differences on a few points (oracle gradient check)
    pts = g['points'].astype(np.float64)

    def f(p):
        return ON.feat_corr_loss(p, counts, gt['feat'], gt['cam'], gt['feat_src'], gt['src_cams'], gt['size'][0], gt['center'][0])
    eps = 1e-6
    for i in range(12):
        for c in range(3):
            p1, p2 = pts.copy(), pts.copy()
            p1[i, c] += eps
            p2[i, c] -= eps
            fd = (f(p1) - f(p2)) / (2 * eps)
            assert abs(fd - g['dpoints'][i, c]) < 2e-3 * max(1.0, abs(g['dpoints'][i, c])) + 2e-6, (i, c, fd, g['dpoints'][i, c])


def test_dsurf_unprojection_vs_reference_golden():
    """oracle_np.dsurf_unproject == the reference's idx_img2cam / idx_cam2world chain (idr.py:234-238) on every pixel."""
    g = golden('dsurf_unproject')
    d = g['depths'].reshape(-1, *g['depths'].shape[-2:])
    pts, valid = ON.dsurf_unproject(d, g['depth_cams'].reshape(-1, 2, 4, 4), g['size'][:1], g['center'][:1])
    assert np.array_equal(valid, g['valid'])
    ref = g['pts_norm']
    assert np.abs(pts[valid] - ref[valid]).max() < 2e-5 * max(1.0, np.abs(ref[valid]).max())
    inb = (np.abs(pts) < float(g['bb'])).all(-1) & valid
    assert (inb != g['inbound']).sum() <= 2                                     # only points within fp32 rounding of the box face may differ


@pytest.mark.parametrize('name', ['feat_corr_v4', 'feat_corr_v8'])
def test_feat_corr_more_source_views(name):
    """V = 4 (the bench configuration) and V = 8 (c3 / c5), B = 8 views of which one has no hit at all."""
    g = golden(name)
    B, P, V = int(g['B']), int(g['P']), int(g['V'])
    _, gt = synth.make_batch(B, P, V, seed=int(g['seed']), size=float(g['scene_size']), center=tuple(g['scene_center']),
                             feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    counts = g['hits'].reshape(B, P).sum(1)
    assert (counts == 0).any()
    loss = ON.feat_corr_loss(g['points'], counts, gt['feat'], gt['cam'], gt['feat_src'], gt['src_cams'], gt['size'][0], gt['center'][0])
    assert abs(loss - float(g['loss'])) < 2e-6


@pytest.mark.parametrize('name', ['carve', 'carve_invalid'])
def test_carving_and_depth_loss_vs_reference_golden(name):
    """oracle_np.carving_t2 / depth_loss == the reference's carving_t2 + get_depth_loss (my_utils.py:269-331, loss.py:37-63) on bumpy
    depth maps with holes: inside / outside voting across views, points no view sees, both attenuation classes."""
    g = golden(name)                                           # carve_invalid: carving_t (conf.use_invalid, my_utils.py:204-266)
    ui = bool(int(g['use_invalid'])) if 'use_invalid' in g.files else False
    size, center = float(g['size'][0]), g['center'][0]
    pw = g['points'].astype(np.float64) / 2 * size + center.astype(np.float64)
    dist, occ, valid = ON.carving_t2(pw, g['depths'][:, 0, 0].astype(np.float64), g['depth_cams'][:, 0].astype(np.float64), use_invalid=ui)
    assert 0.1 < valid.mean() < 0.95 and 0.2 < occ.mean() < 0.8 and (~valid).sum() > 100          # every branch is populated
    bad = (valid != g['in_range']) | (occ != g['occ'])
    assert bad.sum() <= 4, bad.sum()                              # fp32 (reference) vs float64 decisions at pixel / 0.99-depth boundaries
    ok = ~bad
    assert np.abs(dist[ok] - g['dist'][ok]).max() < 2e-5 * max(1.0, np.abs(g['dist'][ok & valid]).max())
    for tag in 'abc':
        fa, na = g['att_' + tag]
        loss, dist_r, w = ON.depth_loss(g['points'], g['eik_out'][0], g['depths'], g['depth_cams'], size, center, 0.25, fa, 0.1, na, use_invalid=ui)
        assert abs(loss - float(g["loss_" + tag])) < 2e-6 * float(g["loss_" + tag]), (tag, loss, float(g["loss_" + tag]))


@pytest.mark.parametrize('name', ['sdf_bwd_w64_skips36', 'sdf_bwd_w64_skip8', 'sdf_bwd_w64_skips48'])
def test_several_skip_connections_vs_reference_golden(name):
    """skip_in = (3, 6) (idr.py:46,86: every listed layer takes cat([x, PE]) / sqrt(2)), (8,) = a skip into the LAST Linear (idr.py:46-49: the
    layer before it is W - d0 wide, u_L = W_L[0, :] splits like any skip layer's adjoint) and (4, 8): value, normal and the double backward of the
    numpy oracle, and the value of the C oracle, against the reference's outputs."""
    from oracle import oracle as O
    g = golden(name)
    skips = tuple(int(v) for v in g['skip_in'])
    sd = synth.make_state_dict(int(g['W']), int(g['seed']), skip_in=skips)
    np.testing.assert_allclose(synth.state_checksum(sd), g['checksum'], rtol=0, atol=0)
    for sk in skips:                                                                                       # out = W - d0 before a skip
        assert sd['implicit_network.lin%d.weight_v' % (sk - 1)].shape == (64 - 39, 64 if (sk - 1) not in skips else 64)
        assert sd['implicit_network.lin%d.weight_v' % sk].shape[1] == 64
    net = ON.sdf_net(sd, skip_in=skips)
    y, n, cache = ON.sdf_forward(net, g['x'])
    np.testing.assert_allclose(y, g['out'], rtol=1e-4, atol=3e-6)
    assert _rel(n, g['grad']) < 2e-5
    dW, db, dx = ON.sdf_backward(net, cache, g['dy'], g['dn'])
    assert _rel(dx, g['dx']) < 1e-4
    for l in range(net.n_layers):
        dv, dg = ON.fold_backward(net.v[l], net.g[l], dW[l])
        assert _rel(dv, g['d_lin%d.weight_v' % l]) < 2e-4, l
        assert _rel(dg, g['d_lin%d.weight_g' % l]) < 2e-4, l
        assert _rel(db[l], g['d_lin%d.bias' % l]) < 2e-4, l
    _, _, dx1 = ON.sdf_backward(net, cache, g['dy'], None)
    assert _rel(dx1, g['dx_value_only']) < 1e-4
    yc = O.sdf_forward(O.Net(sd, skip_in=skips), g['x'])
    np.testing.assert_allclose(yc, g['out'], rtol=1e-4, atol=3e-6)


def test_smooth_depth_term_vs_reference_golden():
    """conf.smooth = 0.05 (loss.py:57-58: SmoothL1(eikonal_output / s, -dist_r / s) * s instead of L1; None in the shipped conf, reachable through IDR_CONF):
    the numpy oracle's depth term on the reference's own step outputs against the reference's depth_loss (fixture idr_w64_smooth), and the L1 form against
    the twin fixture without it."""
    for name in ('idr_w64_smooth', 'idr_w64_tp03'):
        g = golden(name)
        B, P, V, seed = int(g['B']), int(g['P']), int(g['V']), int(g['seed'])
        _, gt = synth.make_batch(B, P, V, seed=seed, size=float(g['scene_size']), center=tuple(g['scene_center']),
                                 feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
        smooth = float(g['smooth']) if 'smooth' in g.files else None
        size, center = float(g['scene_size']), np.asarray(g['scene_center'], np.float64)
        # the fixture holds eikonal_points_hom AFTER the loss rescaled it to world coordinates in place (loss.py:38,42): back to normalised
        pts = (g['out_eikonal_points_hom'][0, :, :3, 0].astype(np.float64) - center) / size * 2
        tp = float(g['tp'])
        near_att = 1 if tp < 1 / 6 else (0.1 if tp < 0.5 else 0.01)
        loss, _, _ = ON.depth_loss(pts, g['out_eikonal_output'].reshape(-1), gt['depths'], gt['depth_cams'], size, center, 0.25, 1, 0.1, near_att, smooth=smooth)
        assert abs(loss - float(g['loss_depth_loss'])) < 2e-5 * max(1.0, float(g['loss_depth_loss'])), (name, loss, float(g['loss_depth_loss']))
    a, b = golden('idr_w64_smooth'), golden('idr_w64_tp03')
    assert float(a['loss_depth_loss']) < float(b['loss_depth_loss']) and np.array_equal(a['out_network_object_mask'], b['out_network_object_mask'])


@pytest.mark.parametrize('name', ['feat_corr', 'feat_corr_v4', 'feat_corr_v8'])
def test_feat_corr_analytic_gradient_vs_reference_autograd(name):
    """oracle_np.feat_corr_loss(with_grad=True) -- projection jacobian x bilinear-tap derivatives x derivative of the normalised correlation -- against the
    d loss / d points the reference's autograd produced (fixture `dpoints`), V = 3 / 4 / 8; and against central differences on a few points."""
    g = golden(name)
    B, P, V = int(g['B']), int(g['P']), int(g['V'])
    _, gt = synth.make_batch(B, P, V, seed=int(g['seed']), size=float(g['scene_size']), center=tuple(g['scene_center']),
                             feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    counts = g['hits'].reshape(B, -1).sum(1)
    args = (counts, gt['feat'], gt['cam'], gt['feat_src'], gt['src_cams'], gt['size'][0], gt['center'][0])
    loss, gr = ON.feat_corr_loss(g['points'], *args, with_grad=True)
    assert abs(loss - float(g['loss'])) < 2e-6
    assert np.abs(gr - g['dpoints']).max() < 2e-5 * np.abs(g['dpoints']).max()
    if name == 'feat_corr':                                                        # central differences on the first view's first points (O(N) loss evaluations)
        n = 6
        c1 = counts.copy()
        c1[1:] = 0
        c1[0] = n
        a1 = (c1,) + args[1:]
        _, ga = ON.feat_corr_loss(g['points'][:n], *a1, with_grad=True)
        _, gf = ON.feat_corr_loss(g['points'][:n], *a1, with_grad='fd')
        assert np.abs(ga - gf).max() < 1e-4 * max(np.abs(gf).max(), 1e-6)
