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


def test_dsurf_sampler_vs_oracle_and_distribution():
    """Phase-0 depth-surface sampling (csrc/sample_kernels.hip, idr.py:226-247).  Exact per-point parity with the oracle's unprojection
    (pinned to the reference golden); the selection is RNG-driven: it must be a duplicate-free, sorted subset of the valid in-box
    pixels, uniform over that set (chi-square over 16 buckets), and different for different seeds."""
    import numpy as np
    from conftest import golden
    from helpers import t
    from oracle import oracle_np as ON
    g = golden('dsurf_unproject')
    depths = g['depths'].reshape(-1, *g['depths'].shape[-2:])
    cams = g['depth_cams'].reshape(-1, 2, 4, 4)
    ref, valid = ON.dsurf_unproject(depths, cams, g['size'][:1], g['center'][:1])
    ref = ref.reshape(-1, 3)
    bb, jr = float(g['bb']), 0.1
    inb = ((np.abs(ref) < bb).all(-1) & valid.reshape(-1))
    n = 256
    seen = np.zeros(ref.shape[0])
    for seed in range(40):
        on, jit, counts, idx = ops.dsurf_samples(t(depths), t(cams), t(g['size'][:1]), t(g['center'][:1]), bb, jr, 1234567 + seed, n)
        assert counts.tolist() == [n, n]
        idx = idx.cpu().numpy()
        for s in range(2):
            assert np.all(np.diff(idx[s]) > 0)                                   # sorted, no duplicates  (np.sort of a choice without replacement)
        assert inb[idx[0]].all()                                                 # on-surface set: valid + inside the box
        assert np.abs(on.cpu().numpy() - ref[idx[0]]).max() < 2e-5              # the points ARE the reference unprojection of those pixels
        pj = jit.cpu().numpy()
        assert valid.reshape(-1)[idx[1]].all() and (np.abs(pj) < bb).all()
        dj = pj - ref[idx[1]]
        assert np.abs(dj).max() <= jr + 1e-5 and np.abs(dj).max() > 0.8 * jr      # jitter in [-0.1, 0.1)^3 (idr.py:239)
        seen[idx[0]] += 1
    # uniformity over the in-box set: 40 draws x 256 of ~1240 pixels
    pool = np.nonzero(inb)[0]
    buckets = np.array_split(seen[pool], 16)
    obs = np.array([b.sum() for b in buckets]); exp = np.array([len(b) for b in buckets]) * (40 * n / len(pool))
    assert ((obs - exp) ** 2 / exp).sum() < 45.0                                 # chi2(15 dof): p ~ 1e-4 at 45
    assert seen[~inb].sum() == 0 and (seen[pool] > 0).mean() > 0.99
    # fewer valid pixels than requested: counts report it (the model raises like np.random.choice does)
    few = depths.copy(); few[:] = 0.0; few[0, 3, 4:9] = depths.max()
    _, _, c2, _ = ops.dsurf_samples(t(few), t(cams), t(g['size'][:1]), t(g['center'][:1]), 1e3, jr, 5, n)
    assert c2.tolist() == [5, 5]
