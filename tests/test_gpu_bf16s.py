"""BASELINE configs[4] ("bf16 MLP weights") -- the mode that is fast AND parity-checked: bf16 weights on the bf16 MFMA, every activation
carried as 2 / 3 bf16 TERMS (csrc/tile_engine_bf16s.h; IDRNetwork.set_trace_dtype('bf16x2' / 'bf16x3')).

The oracle to match is the reference arithmetic on the bf16-rounded weights, `oracle.Net(sd, bf16='weights')` (idr.py:77-94 inside
ray_tracing.py:27-98): the fp32 engine on rounded weights ('bf16w') equals it bit for bit (tests/test_gpu_bf16.py), this engine equals it up
to the order of the fp32 additions inside the matrix core (x3: all 24 mantissa bits of every activation are multiplied) or up to 2^-17 per
activation (x2).  Asserted here: SDF values within a few 1e-7 / 1e-5 of that oracle; tracer hit masks IDENTICAL except rays whose recorded
decision margin (min |sdf|, min |sdf - threshold| over the ray's evaluations, from the oracle's own values) is below 1e-6 (printed); hit depths within
1e-4 relative; at the c5 share the whole training step against the bit-exact 'bf16w' step: losses within 2e-4."""
import numpy as np
import pytest
import torch

from conftest import golden
from helpers import sdf_packed_net, t, trace_params
from mvsdf_amd import build, ops
from mvsdf_amd.utils import synth

pytestmark = pytest.mark.gpu

SDF_TOL = {2: (4e-5, 4e-6), 3: (5e-6, 6e-7)}          # (max, mean) |sdf - oracle| on |sdf| up to ~1.5.  Measured at W = 256: x2 1.1e-5 / 1.7e-6, x3 1.9e-6 / 2.9e-7 (the
# reference's own MKL summation order is 2e-6 away from the oracle: tests/test_oracle_golden.py); profiles/r04_parity_measured.txt
TIE = 1e-6


def _net(sd, terms):
    return ops.pack_bf16_net(sdf_packed_net(sd), terms=terms)


@pytest.mark.parametrize('terms', [2, 3])
@pytest.mark.parametrize('W', [64, 256, 512])
def test_split_mlp_vs_oracle_on_rounded_weights(oracle, W, terms):
    sd = synth.make_state_dict(W, 0)
    net = _net(sd, terms)
    rs = np.random.RandomState(3)
    x = rs.uniform(-1.2, 1.2, size=(4000, 3)).astype(np.float32)
    ref = oracle.sdf_forward(oracle.Net(sd, bf16='weights'), x, ncols=1)[:, 0]
    ys = {}
    for mt in (1, 2, 4):
        y = ys[mt] = ops.sdf_col0(net, t(x), mt=mt).cpu().numpy()
        d = np.abs(y - ref)
        print('W=%d x%d mt=%d: |sdf - oracle(bf16 weights)| max %.3g mean %.3g' % (W, terms, mt, d.max(), d.mean()))
        assert d.max() < SDF_TOL[terms][0] and d.mean() < SDF_TOL[terms][1]
    assert np.array_equal(ys[1], ys[2]) and np.array_equal(ys[1], ys[4])        # row tiling does not change a row's arithmetic


_CARRY_SCRIPT = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import sdf_packed_net, t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
x = np.random.RandomState(3).uniform(-1.2, 1.2, size=(3000, 3)).astype(np.float32)
out = {}
for W in (64, 256, 512):
    for terms in (2, 3):
        net = ops.pack_bf16_net(sdf_packed_net(synth.make_state_dict(W, 0)), terms=terms)
        for mt in (1, 2):
            out['%%d_%%d_%%d' %% (W, terms, mt)] = ops.sdf_col0(net, t(x), mt=mt).cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_the_two_weight_fetch_schemes_of_the_split_engine_agree_bit_for_bit(tmp_path):
    """ROLLING (row-sample kernels) and CARRIED (k_sphere_trace) fetch the same weights for the same matrix instructions in the same order."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for c in ('0', '1'):
        f = str(tmp_path / ('y%s.npz' % c))
        subprocess.check_call([sys.executable, '-c', _CARRY_SCRIPT % (root, os.path.join(root, 'tests')), f], env=dict(os.environ, MVSDF_BF_CARRY=c, MVSDF_LIB=build.build(tag='dev')))   # a development switch: the dev build of the library reads it
        res.append(np.load(f))
    assert set(res[0].files) == set(res[1].files) and len(res[0].files) == 12
    for k in res[0].files:
        assert np.array_equal(res[0][k], res[1][k]), k


def _compare_with_oracle(tag, oracle, sd, W, cam, dirs, om, training, steps, terms, mt, mt_samples, net=None, onet=None):
    """tracer on the split engine vs oracle.trace on the rounded weights (with the oracle's per-ray decision margins)
    net / onet: another packed network / oracle network (tests/test_gpu_f32x3.py: the three-term weights against the fp32 oracle)"""
    net = _net(sd, terms) if net is None else net
    B, P = dirs.shape[:2]
    iv = torch.linspace(0, 1, 100)
    pts, mask, dists, cnt = ops.trace(net, t(cam), t(dirs), t(om), trace_params(W), training, iv.cuda(), t(steps), mt=mt, mt_samples=mt_samples)
    mask, dists, cnt = mask.cpu().numpy(), dists.cpu().numpy(), cnt.cpu().numpy()
    p_o, m_o, d_o, rows, mg = oracle.trace(oracle.Net(sd, bf16='weights') if onet is None else onet, cam, dirs, om, training, steps, iv.numpy(), margins=True,
                                           **synth.model_conf(W)['ray_tracer'])
    margin = mg.min(axis=1)
    diff = np.nonzero(mask != m_o)[0]
    both = mask & m_o
    rel = np.abs(dists - d_o) / np.maximum(np.abs(d_o), 1e-12)
    print('%s x%d: masks differ on %d of %d rays%s; hit depth rel max %.3g p99 %.3g median %.3g; rows %s vs oracle %s' % (
        tag, terms, diff.size, mask.size, ''.join(' [ray %d margin %.3g]' % (i, margin[i]) for i in diff[:8]),
        rel[both].max(), np.percentile(rel[both], 99), np.median(rel[both]), cnt[:4].tolist(), rows.tolist()))
    # hit masks identical except at recorded ties
    assert all(margin[i] < TIE for i in diff), [(int(i), float(margin[i])) for i in diff]
    # depths of the hit rays within 1e-4 relative (north_star), ties excepted: a ray within 1e-6 of a decision may take the other branch
    worst = int(np.argmax(np.where(both, rel, 0)))
    print('   largest depth difference: ray %d rel %.3g, its decision margin %.3g' % (worst, rel[worst], margin[worst]))
    bad = np.nonzero(both & (rel > 1e-4))[0]
    print('   rays beyond 1e-4: %d%s' % (bad.size, ''.join(' [ray %d rel %.3g margin %.3g]' % (i, rel[i], margin[i]) for i in bad[:8])))
    assert all(margin[i] < TIE for i in bad), [(int(i), float(rel[i]), float(margin[i])) for i in bad]
    # the row counters (evaluations per stage) move only with such ties
    assert np.abs(cnt[:4].astype(np.int64) - rows).max() <= 200 * (diff.size + bad.size) + max(400, rows.max() // 100)     # (a sampler ray more or less = 100 rows)
    return mask, dists, diff.size


@pytest.mark.parametrize('terms', [2, 3])
@pytest.mark.parametrize('W,mode', [(64, 'train'), (256, 'eval'), (256, 'train'), (512, 'train')])
def test_split_tracer_vs_oracle_on_rounded_weights(oracle, W, mode, terms):
    g = golden('trace_mlp_w%d_%s' % (W, mode))
    sd = synth.make_state_dict(W, int(g['seed']))
    B, P = int(g['B']), int(g['P'])
    dirs = g['ray_dirs'].reshape(B, P, 3)
    mask, dists, _ = _compare_with_oracle('trace_mlp_w%d_%s' % (W, mode), oracle, sd, W, g['cam_loc'], dirs, np.ones(B * P, bool), mode == 'train',
                                          g['minsdf_steps'], terms, 1, 2)
    # all chunkings give the same result (rows are independent)
    net = _net(sd, terms)
    iv = torch.linspace(0, 1, 100)
    p2, m2, d2, _ = ops.trace(net, t(g['cam_loc']), t(dirs), torch.ones(B * P, dtype=torch.bool, device='cuda'), trace_params(W), mode == 'train', iv.cuda(),
                              t(g['minsdf_steps']), mt=2, mt_samples=4)
    assert np.array_equal(m2.cpu().numpy(), mask) and np.array_equal(d2.cpu().numpy(), dists)
    # and the accuracy budget of "bf16 MLP weights" itself against the fp32 reference (the weights' 8-bit mantissas; the same as 'bf16w')
    agree_r = (mask == g['mask']).mean()
    both = mask & g['mask']
    rel_r = np.abs(dists - g['dists'])[both] / np.abs(g['dists'][both])
    print('   vs fp32 reference: masks agree %.4f, hit depth rel p99 %.3g' % (agree_r, np.percentile(rel_r, 99)))
    assert agree_r >= 0.98 and np.percentile(rel_r, 99) < 5e-3


@pytest.mark.parametrize('terms', [2, 3])
def test_split_tracer_at_the_c5_share_vs_oracle(oracle, terms):
    """One GPU's share of BASELINE configs[4] (4096 rays of the idr_c5share scene, 8x256 MLP, training mode, the object mask of the batch)."""
    g = golden('idr_c5share')
    W, B, P, V, seed = int(g['W']), int(g['B']), int(g['P']), int(g['V']), int(g['seed'])
    assert (W, B * P) == (256, 4096)
    sd = synth.make_state_dict(W, seed)
    inp, _ = synth.make_batch(B, P, V, seed=seed, size=float(g['scene_size']), center=tuple(g['scene_center']),
                              feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    dirs, cam = ops.camera_rays(t(inp['uv']), t(inp['pose']), t(inp['intrinsics']))
    steps = np.random.RandomState(seed).uniform(size=100).astype(np.float32)
    om = np.asarray(inp['object_mask']).reshape(-1).astype(bool)
    _compare_with_oracle('c5 share', oracle, sd, W, cam.cpu().numpy(), dirs.cpu().numpy(), om, True, steps, terms, 2, 2)


@pytest.mark.parametrize('dtype', ['bf16x2', 'bf16x3'])
def test_split_step_at_the_c5_share_vs_the_bit_exact_weights_only_step(dtype):
    """The whole training step at the c5 share with the split tracer against the same step with the 'bf16w' tracer (bit-exact vs the oracle on the
    rounded weights): hit masks (count of differing rays printed, <= 2 of 4096: ties), hit depths 1e-4, every loss term 2e-4, gradient norm 1e-3.
    And against the fp32 reference fixture: the budget of rounding the weights, the same as 'bf16w' (depth p99 ~1.2e-3)."""
    from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
    from mvsdf_amd.model.loss import IDRLoss
    from mvsdf_amd.utils.config import ConfigDict
    g = golden('idr_c5share')
    W, B, P, V, seed, tp = int(g['W']), int(g['B']), int(g['P']), int(g['V']), int(g['seed']), float(g['tp'])
    inp, gt = synth.make_batch(B, P, V, seed=seed, size=float(g['scene_size']), center=tuple(g['scene_center']),
                               feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    cam = np.repeat(inp['pose'][:, :3, 3], P, axis=0)
    res = {}
    for dt in ('bf16w', dtype):
        m = IDRNetwork(ConfigDict(synth.model_conf(W)))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, seed).items()})
        m = m.cuda().train().set_trace_dtype(dt)
        torch.manual_seed(seed + 5)
        out = m({k: t(v) for k, v in inp.items()}, tp)
        lo = IDRLoss()(out, {k: t(v) for k, v in gt.items()}, tp, B)
        lo['loss'].backward()
        gn = float(torch.cat([p.grad.flatten() for p in m.parameters()]).norm())
        mask = out['network_object_mask'].cpu().numpy()
        depth = np.linalg.norm(out['points'].detach().cpu().numpy() - cam, axis=1)
        res[dt] = (mask, depth, {k: float(lo[k].detach().reshape(-1)[0]) for k in ('loss', 'rgb_loss', 'eikonal_loss', 'depth_loss', 'feat_loss', 'surf_loss')}, gn)
    (m0, d0, l0, g0), (m1, d1, l1, g1) = res['bf16w'], res[dtype]
    ndiff = int((m0 != m1).sum())
    both = m0 & m1
    rel = np.abs(d1 - d0)[both] / d0[both]
    dl = max(abs(l1[k] - l0[k]) / max(abs(l0[k]), 1e-3) for k in l0)
    print('c5 share step, %s vs bf16w: masks differ on %d rays, hit depth rel max %.3g p99 %.3g, %d rays beyond 1e-4, worst loss term off by %.3g, |grad| %.6g vs %.6g' % (
        dtype, ndiff, rel.max(), np.percentile(rel, 99), int((rel > 1e-4).sum()), dl, g1, g0))
    assert ndiff <= 2 and int((rel > 1e-4).sum()) <= 2 and np.percentile(rel, 99) < (5e-5 if dtype == 'bf16x2' else 5e-6)     # measured: 1.2e-5 / 7e-7
    assert dl <= 2e-4 and abs(g1 - g0) <= 1e-3 * g0
    mref = g['out_network_object_mask']
    dref = np.linalg.norm(g['out_points'] - cam, axis=1)
    b2 = m1 & mref
    rr = np.abs(d1 - dref)[b2] / dref[b2]
    print('   vs the fp32 reference fixture: masks agree %.4f, depth rel p99 %.3g max %.3g' % (float((m1 == mref).mean()), np.percentile(rr, 99), rr.max()))
    assert (m1 == mref).mean() >= 0.995 and np.percentile(rr, 99) < 2e-3
