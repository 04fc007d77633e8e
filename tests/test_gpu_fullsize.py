"""Parity at BASELINE.json's full sizes (c2: 2048 rays, W=256; c3-shaped: 8192 rays) -- against the oracle where it finishes in
seconds, and through size-independent properties of the domain otherwise."""
import numpy as np
import pytest
import torch

from helpers import sdf_packed_net, t, trace_params
from mvsdf_amd import ops
from mvsdf_amd.utils import synth

pytestmark = pytest.mark.gpu


def _scene(B, P, W=256, seed=0):
    sd = synth.make_state_dict(W, 0)
    inp, _ = synth.make_batch(B, P, 0, seed=seed, with_features=False)
    net = sdf_packed_net(sd)
    dirs, cam = ops.camera_rays(t(inp['uv']), t(inp['pose']), t(inp['intrinsics']))
    return sd, net, dirs, cam


@pytest.mark.parametrize('B,P,mt,name,dtype', [(8, 256, 1, 'c2', 'f32x3'), (8, 512, 2, 'c5 share', 'f32x3'), (8, 1024, 4, 'c3', 'f32x3'),
                                               (8, 256, 1, 'c2', 'f32'), (8, 512, 2, 'c5 share', 'f32'), (8, 1024, 4, 'c3', 'f32'), (8, 4096, 4, 'c5 whole batch', 'f32')])
def test_full_size_tracer_bit_exact_vs_oracle(oracle, B, P, mt, name, dtype):
    """BASELINE's batches -- c2: 2048 rays, c5's per-GPU share: 4096, c3: 8192, the whole c5 batch: 32768 -- through the 8x256 MLP in training mode with the row-tile
    settings RayTracing picks at that size: masks, dists, points and the per-stage row counters equal the CPU oracle bit for bit, for the product's default
    tracing arithmetic 'f32x3' (oracle: the model of v_mfma_f32_16x16x32_bf16, eight columns per AVX2 instruction: ~2.6 ms per MLP row and host thread, c3's
    585 k rows ~15 s on the GPU box's 128 threads) and for the fmaf-chain arithmetic 'f32' (c3 about a second, the whole c5 batch's 2.3 M rows a few)."""
    sd, net, dirs, cam = _scene(B, P)
    if dtype == 'f32x3':
        net = ops.pack_bf16_net(net, terms=3, weight_terms=3)
    onet = oracle.Net(sd, bf16='f32x3' if dtype == 'f32x3' else False)
    R = B * P
    iv = torch.linspace(0, 1, 100)
    steps = np.random.RandomState(1).uniform(size=100).astype(np.float32)
    om = np.ones(R, bool)
    pts, mask, dists, cnt = ops.trace(net, cam, dirs, t(om), trace_params(256), True, iv.cuda(), t(steps), mt=mt, mt_samples=2)
    p_o, m_o, d_o, rows = oracle.trace(onet, cam.cpu().numpy(), dirs.cpu().numpy(), om, True, steps, iv.numpy(), **synth.model_conf(256)['ray_tracer'])
    assert np.array_equal(mask.cpu().numpy(), m_o) and np.array_equal(dists.cpu().numpy(), d_o) and np.array_equal(pts.cpu().numpy(), p_o)
    assert np.array_equal(cnt.cpu().numpy()[:4], rows)
    assert 0.5 < m_o.mean() < 0.95


@pytest.mark.parametrize('B,P', [(8, 256), (8, 1024)])
def test_tracer_properties_full_size(B, P):
    """Size-independent properties: (1) launch geometry (rays per workgroup, list chunking) never changes a bit; (2) rays are
    independent: tracing a permuted / split batch gives the same per-ray results; (3) hit points lie on the surface (|sdf| small),
    inside the bounding sphere, dists within the sphere-intersection interval; (4) eval and train agree on every non-min-sdf ray."""
    sd, net, dirs, cam = _scene(B, P)
    R = B * P
    iv = torch.linspace(0, 1, 100).cuda()
    torch.manual_seed(0)
    steps = torch.empty(100).uniform_(0, 1).cuda()
    om = torch.ones(R, dtype=torch.bool, device='cuda')
    tp = trace_params(256)
    ref = ops.trace(net, cam, dirs, om, tp, True, iv, steps, mt=1, mt_samples=2)
    for mt, rpw in ((2, 4), (4, 8), (2, 3)):
        o = ops.trace(net, cam, dirs, om, tp, True, iv, steps, mt=mt, mt_samples=rpw)
        assert torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2]) and torch.equal(o[0], ref[0]) and torch.equal(o[3][:4], ref[3][:4])
    # (2) per-view split: each view traced alone
    for b in (0, B - 1):
        o = ops.trace(net, cam[b:b + 1].contiguous(), dirs[b:b + 1].contiguous(), om[:P], tp, True, iv, steps)
        assert torch.equal(o[1], ref[1][b * P:(b + 1) * P]) and torch.equal(o[2], ref[2][b * P:(b + 1) * P])
    # (3) geometry
    pts, mask, dists = ref[0], ref[1], ref[2]
    sdf = ops.sdf_col0(net, pts[mask].contiguous())
    assert float(sdf.abs().max()) < 5e-3 and float(sdf.abs().median()) < 1e-4
    assert float(pts[mask].norm(dim=1).max()) <= 1.0 + 1e-5
    tt, isect = ops.sphere_intersection(cam, dirs)
    tt = tt.reshape(-1, 2)
    assert bool((dists[mask] >= tt[mask][:, 0] - 1e-6).all()) and bool((dists[mask] <= tt[mask][:, 1] + 1e-6).all())
    assert not bool(mask[~isect.reshape(-1)].any())
    # (4) eval mode: same masks; same dists except where training ran the min-sdf search / left-out projection
    ev = ops.trace(net, cam, dirs, om, tp, False, iv, None)
    assert torch.equal(ev[1], mask)
    assert torch.equal(ev[2][mask], dists[mask])
    assert int(ev[3][3]) == 0 and int(ref[3][3]) > 0


def test_mlp_linearity_free_property_and_fold_idempotence():
    """weight_norm fold: scaling weight_v by a power of two leaves W bit-identical (direction only); g scales W linearly."""
    rs = np.random.RandomState(0)
    v = torch.from_numpy(rs.normal(size=(256, 256)).astype(np.float32)).cuda()
    g = torch.from_numpy(rs.uniform(0.5, 2, size=(256, 1)).astype(np.float32)).cuda()
    w1, _, _ = ops.fold_pack(v, g)
    w2, _, _ = ops.fold_pack(v * 4.0, g)
    w3, _, _ = ops.fold_pack(v, g * 2.0)
    assert torch.equal(w1, w2) and torch.equal(w3, w1 * 2.0)
    assert torch.allclose(w1.norm(dim=1, keepdim=True), g, rtol=1e-6)
