"""Direct GPU parity of the loss kernels (csrc/loss_kernels.hip) and the stand-alone sphere-intersection kernel against goldens
recorded from the PyTorch reference (tests/golden/make_golden.py): k_feat_corr's loss AND its analytic d/d(points) for V = 3 / 4 / 8
source views in both feature layouts, k_carve on bumpy depth maps with holes, k_sphere_intersection on rays.npz."""
import numpy as np
import pytest
import torch

from conftest import golden
from helpers import t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth

pytestmark = pytest.mark.gpu


def _feat_inputs(g):
    B, P, V = int(g['B']), int(g['P']), int(g['V'])
    _, gt = synth.make_batch(B, P, V, seed=int(g['seed']), size=float(g['scene_size']), center=tuple(g['scene_center']),
                             feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    counts = g['hits'].reshape(B, P).sum(1)
    vs = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).cuda()
    return B, V, gt, vs


@pytest.mark.parametrize('name', ['feat_corr', 'feat_corr_v4', 'feat_corr_v8'])
@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
def test_feat_corr_loss_and_dpoints_vs_reference(name, layout):
    """get_feat_loss_corr (loss.py:115-165): loss 1e-4 rel, d loss / d points 2e-4 of its largest entry (autograd of the reference
    through grid_sample / the projections), per point."""
    g = golden(name)
    B, V, gt, vs = _feat_inputs(g)
    feat, fsrc = t(gt['feat']), t(gt['feat_src'])
    if layout == 'channels_last':                             # same shapes, channel stride 1 (each bilinear tap = one 128-byte line)
        feat = feat.contiguous(memory_format=torch.channels_last)
        fsrc = fsrc.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
        assert feat.stride(1) == 1 and fsrc.stride(2) == 1
    loss_pp, dpts = ops.feat_corr(t(g['points']), vs, feat, fsrc, t(gt['cam']), t(gt['src_cams']), t(gt['size'][:1]), t(gt['center'][:1]))
    loss = float(loss_pp.double().sum())
    assert abs(loss - float(g['loss'])) <= 1e-4 * float(g['loss']), (loss, float(g['loss']))
    ref = g['dpoints']
    err = np.abs(dpts.cpu().numpy() - ref).max()
    assert err <= 2e-4 * np.abs(ref).max(), (err, np.abs(ref).max())
    assert (np.abs(ref).sum(1) > 0).mean() > 0.5              # the fixture's gradient is not trivially zero


@pytest.mark.parametrize('name', ['carve', 'carve_invalid'])
def test_depth_carve_vs_reference(name):
    """carving_t2 (`carve`) / carving_t (`carve_invalid`: conf.use_invalid, loss.py:43-46) + the weighting of get_depth_loss (my_utils.py:204-331, loss.py:37-63) on bumpy
    depth maps with holes and a depth step; the fp32 decisions (nearest pixel, depth > 0.99 * gathered, in-range) may differ from torch's on a handful of boundary points."""
    g = golden(name)
    ui = bool(int(g['use_invalid'])) if 'use_invalid' in g.files else False
    size, center = g['size'][:1], g['center'][:1]
    depths, cams = t(g['depths'][:, 0, 0]), t(g['depth_cams'][:, 0])
    dist_ref = g['dist'] / float(size[0]) * 2 + (-1.25) * (~g['in_range'])
    dist_ref = np.clip(dist_ref, -1.25, 1.25)
    M = dist_ref.shape[0]
    for tag in 'abc':
        fa, na = [float(v) for v in g['att_' + tag]]
        pts = t(g['points'])
        dist_r, w = ops.depth_carve(pts, depths, cams, t(size), t(center), 1 / 8, 0.25, fa, 0.1, na, use_invalid=ui)
        dist_r, w = dist_r.cpu().numpy(), w.cpu().numpy()
        far, near = np.abs(dist_ref) > 0.25, np.abs(dist_ref) < 0.1
        w_ref = (far * fa + ~far) * (near * na + ~near) * g['in_range']
        bad = np.abs(dist_r - dist_ref) > 1e-5
        assert bad.sum() <= 4, (tag, bad.sum())
        assert np.array_equal((w > 0)[~bad], g['in_range'][~bad])
        assert np.abs(w - w_ref)[~bad].max() < 1e-6
        loss = float((np.abs(g['eik_out'][0].astype(np.float64) + dist_r) * w).mean())
        assert abs(loss - float(g['loss_' + tag])) <= 2e-3 * float(g['loss_' + tag]) * max(1, bad.sum()) + 1e-6, (tag, loss, float(g['loss_' + tag]))
    # homogeneous [M,4] points rescaled in place: the side effect of loss.py:38,42
    hom = torch.cat([t(g['points']), torch.ones(M, 1, device='cuda')], 1).contiguous()
    d2, w2 = ops.depth_carve(hom, depths, cams, t(size), t(center), 1 / 8, 0.25, 1.0, 0.1, 1.0, world_inplace=True, use_invalid=ui)
    world = g['points'] / 2 * float(size[0]) + center
    assert np.abs(hom[:, :3].cpu().numpy() - world).max() < 1e-6 and torch.equal(hom[:, 3], torch.ones(M, device='cuda'))
    d1, _ = ops.depth_carve(t(g['points']), depths, cams, t(size), t(center), 1 / 8, 0.25, 1.0, 0.1, 1.0, use_invalid=ui)
    assert torch.equal(d1, d2)


def test_sphere_intersection_vs_reference(oracle):
    """rend_util.get_sphere_intersection (rend_util.py:141-162), the stand-alone kernel: mask equal, t_near / t_far within fp32 rounding of
    torch's (its CPU sqrt is not correctly rounded), bit-exact vs the C oracle."""
    g = golden('rays')
    tt, m = ops.sphere_intersection(t(g['cam_loc']), t(g['ray_dirs']), 1.0)
    assert np.array_equal(m.cpu().numpy(), g['mask_intersect'])
    assert (~g['mask_intersect']).sum() > 0
    ref = g['sphere_intersections']
    assert np.abs(tt.cpu().numpy() - ref).max() <= 4e-7 * max(1.0, np.abs(ref).max())
    assert (tt.cpu().numpy()[~g['mask_intersect']] == 0).all()
    if hasattr(oracle, 'sphere_intersection'):
        to, mo = oracle.sphere_intersection(g['cam_loc'], g['ray_dirs'], 1.0)
        assert np.array_equal(tt.cpu().numpy(), to) and np.array_equal(m.cpu().numpy(), mo)
