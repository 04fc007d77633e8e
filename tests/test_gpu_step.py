"""Bookkeeping kernels of the training step (csrc/step_kernels.hip) vs the torch expressions they replace (idr.py:202-213, 272)."""
import pytest
import torch

from mvsdf_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('R,p_hit,use_om', [(2048, 0.7, False), (3000, 0.3, True), (1, 1.0, False), (1025, 0.0, True), (5000, 1.0, True)])
def test_partition_rays_matches_boolean_mask_order(R, p_hit, use_om):
    g = torch.Generator().manual_seed(R)
    net_mask = (torch.rand(R, generator=g) < p_hit).cuda()
    om = (torch.rand(R, generator=g) < 0.8).cuda() if use_om else None
    tm = (torch.rand(R, generator=g) < 0.6).cuda()
    dirs = torch.randn(R, 3, generator=g).cuda()
    perm, inv, true_rows, counts, view = ops.partition_rays(net_mask, om, tm, dirs)
    surf = net_mask & om if use_om else net_mask
    idx = torch.arange(R, device='cuda')
    ref_perm = torch.cat([idx[surf], idx[~surf]])                                   # boolean-mask order of the reference
    assert torch.equal(perm, ref_perm)
    assert torch.equal(inv[perm], idx)
    N, n_true = int(counts[0]), int(counts[1])
    assert N == int(surf.sum()) and n_true == int((surf & tm).sum())
    assert torch.equal(true_rows[:n_true], torch.nonzero(tm[perm[:N]]).flatten())
    assert torch.equal(view, -dirs[perm])
