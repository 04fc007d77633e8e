"""BASELINE configs[4]: the tracing MLP with bf16 weights / activations on the bf16 MFMA (csrc/tile_engine_bf16.h).

Outside the 1e-4 parity claim.  Two references:
  * the oracle's bf16 twin (oracle_mvsdf.c::sdf_row_bf16): same rounding points, fp32 k-ordered accumulation.  The matrix core sums each
    instruction's 32 products with its own internal alignment, so the kernel is not bit-identical to any CPU model; the difference is
    accumulation noise (~1e-6) amplified where it flips a bf16 rounding of an activation;
  * the fp32 reference goldens: the ACCURACY BUDGET of the variant -- how far 8-bit weight mantissas move the traced surface."""
import numpy as np
import pytest
import torch

from conftest import golden
from helpers import sdf_packed_net, t, trace_params
from mvsdf_amd import ops
from mvsdf_amd.utils import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('W', [64, 256, 512])
def test_bf16_mlp_vs_oracle_twin_and_fp32_reference(oracle, W):
    sd = synth.make_state_dict(W, 0)
    net = sdf_packed_net(sd, bf16=True)
    rs = np.random.RandomState(3)
    x = rs.uniform(-1.2, 1.2, size=(4000, 3)).astype(np.float32)
    ref = oracle.sdf_forward(oracle.Net(sd, bf16=True), x, ncols=1)[:, 0]
    f32 = oracle.sdf_forward(oracle.Net(sd), x, ncols=1)[:, 0]
    for mt in (1, 2, 4):
        y = ops.sdf_col0(net, t(x), mt=mt).cpu().numpy()
        d = np.abs(y - ref)
        print('W=%d mt=%d: vs bf16 twin max %.3g mean %.3g; vs fp32 max %.3g mean %.3g' % (W, mt, d.max(), d.mean(), np.abs(y - f32).max(), np.abs(y - f32).mean()))
        assert d.max() < 6e-3 and d.mean() < 1e-4            # twin: same function up to accumulation noise; a flipped bf16 rounding of one activation moves the output by ~1e-3
        assert np.abs(y - f32).max() < 2e-2 and np.abs(y - f32).mean() < 2e-3   # budget vs the fp32 network (|sdf| up to ~1.5)
    y1 = ops.sdf_col0(net, t(x), mt=1).cpu().numpy()
    assert np.array_equal(y1, ops.sdf_col0(net, t(x), mt=4).cpu().numpy())      # row tiling does not change a row's arithmetic
    if W == 256:
        g = golden('sdf_w256')
        yg = ops.sdf_col0(sdf_packed_net(synth.make_state_dict(256, int(g['seed'])), bf16=True), t(g['x'])).cpu().numpy()
        assert np.abs(yg - g['out'][:, 0]).max() < 2e-2                            # vs the PyTorch reference itself


_CARRY_SCRIPT = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import sdf_packed_net, t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
x = np.random.RandomState(3).uniform(-1.2, 1.2, size=(3000, 3)).astype(np.float32)
out = {}
for W in (64, 256, 512):
    net = sdf_packed_net(synth.make_state_dict(W, 0), bf16=True)
    for mt in (1, 2, 4):
        out['%%d_%%d' %% (W, mt)] = ops.sdf_col0(net, t(x), mt=mt).cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_the_two_weight_fetch_schemes_of_the_bf16_engine_agree_bit_for_bit(tmp_path):
    """tile_engine_bf16.h fetches weights in two ways (ROLLING: the row-sample kernels; CARRIED: k_sphere_trace).  Same arithmetic in the same
    order: identical bits.  MVSDF_BF_CARRY is read once per process, hence the two child processes."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for c in ('0', '1'):
        f = str(tmp_path / ('y%s.npz' % c))
        subprocess.check_call([sys.executable, '-c', _CARRY_SCRIPT % (root, os.path.join(root, 'tests')), f], env=dict(os.environ, MVSDF_BF_CARRY=c))
        res.append(np.load(f))
    assert set(res[0].files) == set(res[1].files) and len(res[0].files) == 9
    for k in res[0].files:
        assert np.array_equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize('W,mode', [(64, 'train'), (256, 'eval'), (256, 'train')])
def test_bf16_tracer_vs_oracle_twin_and_reference_golden(oracle, W, mode):
    g = golden('trace_mlp_w%d_%s' % (W, mode))
    sd = synth.make_state_dict(W, int(g['seed']))
    net = sdf_packed_net(sd, bf16=True)
    B, P = int(g['B']), int(g['P'])
    cam, dirs = t(g['cam_loc']), t(g['ray_dirs']).reshape(B, P, 3)
    om = torch.ones(B * P, dtype=torch.bool, device='cuda')
    training = mode == 'train'
    iv = torch.linspace(0, 1, 100)
    pts, mask, dists, cnt = ops.trace(net, cam, dirs, om, trace_params(W), training, iv.cuda(), t(g['minsdf_steps']), mt=1, mt_samples=2)
    mask, dists = mask.cpu().numpy(), dists.cpu().numpy()
    # (1) the oracle's bf16 twin
    p_o, m_o, d_o, rows = oracle.trace(oracle.Net(sd, bf16=True), g['cam_loc'], g['ray_dirs'], np.ones(B * P, bool), training, g['minsdf_steps'],
                                       iv.numpy(), **synth.model_conf(W)['ray_tracer'])
    agree = (mask == m_o).mean()
    both = mask & m_o
    rel = np.abs(dists - d_o)[both] / np.abs(d_o[both])
    print('W=%d %s vs bf16 twin: masks agree %.4f, hit depth rel max %.3g p99 %.3g median %.3g' % (W, mode, agree, rel.max(), np.percentile(rel, 99), np.median(rel)))
    # measured: W=256 median 1.1e-5, p99 1.15e-3, max 2.5e-3; W=64 median 0, p99 1.1e-3, max 1.4e-2 (the tail = rays where one flipped bf16
    # rounding moves a sphere-tracing step across the 5e-5 threshold)
    assert agree >= 0.995 and np.percentile(rel, 99) < 1.5e-3 and np.median(rel) < 5e-5
    # (2) accuracy budget against the fp32 PyTorch reference
    agree_r = (mask == g['mask']).mean()
    both = mask & g['mask']
    rel_r = np.abs(dists - g['dists'])[both] / np.abs(g['dists'][both])
    print('W=%d %s vs fp32 reference: masks agree %.4f, hit depth rel max %.3g 99%% %.3g median %.3g' % (W, mode, agree_r, rel_r.max(), np.percentile(rel_r, 99), np.median(rel_r)))
    assert agree_r >= 0.98 and np.percentile(rel_r, 99) < 5e-3 and np.median(rel_r) < 5e-4
    # the TAIL, not only percentiles: a regression that doubles the outliers must fail.  Measured: W=256 max 1.6e-3, none beyond 1e-2;
    # W=64 train max 7.3e-2, 7 of 3419 rays beyond 1e-2 (the narrow net's surface has thin features the 8-bit mantissas miss)
    n_bad = int((rel_r > 1e-2).sum())
    print('   tail: max %.3g, rays with depth error > 1e-2: %d of %d' % (rel_r.max(), n_bad, rel_r.size))
    assert rel_r.max() < (0.15 if W == 64 else 1e-2) and n_bad <= (12 if W == 64 else 0)
    # all chunkings give the same result (rows are independent)
    p2, m2, d2, _ = ops.trace(net, cam, dirs, om, trace_params(W), training, iv.cuda(), t(g['minsdf_steps']), mt=2, mt_samples=4)
    assert np.array_equal(m2.cpu().numpy(), mask) and np.array_equal(d2.cpu().numpy(), dists)


def test_bf16_training_step_runs_and_stays_close_to_fp32():
    """IDRNetwork.set_trace_dtype('bf16'): the tracer runs in bf16, the differentiable passes in fp32.  Same batch, fp32 vs bf16 tracer: the
    hit sets agree to >= 98 % and the losses stay within a few per cent."""
    from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
    from mvsdf_amd.model.loss import IDRLoss
    from mvsdf_amd.utils.config import ConfigDict
    W = 256
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0).items()}
    inp, gt = synth.make_batch(4, 256, 4, seed=2, feat_hw=(60, 80))
    inp, gt = {k: t(v) for k, v in inp.items()}, {k: t(v) for k, v in gt.items()}
    res = {}
    for dt in ('f32', 'bf16'):
        m = IDRNetwork(ConfigDict(synth.model_conf(W)))
        m.load_state_dict(sd)
        m = m.cuda().train().set_trace_dtype(dt)
        torch.manual_seed(0)
        out = m(inp, 0.3)
        lo = IDRLoss()(out, dict(gt), 0.3, 4)
        lo['loss'].backward()
        gn = torch.cat([p.grad.flatten() for p in m.parameters()]).norm()
        res[dt] = (out['network_object_mask'].clone(), {k: float(v.detach()) for k, v in lo.items()}, float(gn))
        assert torch.isfinite(gn)
    agree = (res['f32'][0] == res['bf16'][0]).float().mean().item()
    print('hit masks agree %.4f; losses f32 %s bf16 %s' % (agree, res['f32'][1], res['bf16'][1]))
    assert agree >= 0.98
    for k in ('loss', 'rgb_loss', 'eikonal_loss', 'depth_loss'):
        assert abs(res['bf16'][1][k] - res['f32'][1][k]) <= 0.05 * max(abs(res['f32'][1][k]), 1e-3), k


def test_bf16_tracer_with_parameters_in_the_flat_optimizer_buffer():
    """FlatAdam moves every parameter into one flat buffer: the bias vectors then start at arbitrary 4-byte offsets (layer 3 has 217 outputs), and
    the bf16 engine reads them with 16-byte loads up to the next multiple of 16 entries (into the next parameter).  Same outputs as with
    separately allocated parameters, bit for bit."""
    from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
    from mvsdf_amd.optim import FlatAdam
    from mvsdf_amd.utils.config import ConfigDict
    W = 256
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0).items()}
    inp, _ = synth.make_batch(2, 256, 2, seed=2, feat_hw=(60, 80))
    inp = {k: t(v) for k, v in inp.items()}
    res = []
    for flat in (False, True):
        m = IDRNetwork(ConfigDict(synth.model_conf(W)))
        m.load_state_dict(sd)
        m = m.cuda().train().set_trace_dtype('bf16')
        if flat:
            FlatAdam(m.parameters(), lr=0.0)
            offs = sorted(p.data_ptr() % 16 for n, p in m.named_parameters() if n.endswith('bias'))
            assert offs[0] != offs[-1] or offs[0] != 0, 'expected biases at unaligned offsets of the flat buffer'
        torch.manual_seed(0)
        out = m(inp, 0.3)
        res.append({k: out[k].detach().clone() for k in ('network_object_mask', 'points', 'rgb_values')})
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


def test_bf16_step_at_the_c5_per_gpu_shape_vs_the_fp32_reference():
    """BASELINE configs[4] (32768 rays, V = 8, bf16 MLP weights over 8 GPUs) = 4096 rays per GPU: the bf16-tracer step on that share against
    the fp32 REFERENCE fixture idr_c5share -- the accuracy budget of the mode (outside the 1e-4 claim), measured and asserted:
    hit masks agree on >= 99.5 % of the rays, hit depths within 2e-3 relative at the 99th percentile, every loss term within 3 %."""
    from conftest import golden
    from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
    from mvsdf_amd.model.loss import IDRLoss
    from mvsdf_amd.utils.config import ConfigDict
    g = golden('idr_c5share')
    W, B, P, V, seed, tp = int(g['W']), int(g['B']), int(g['P']), int(g['V']), int(g['seed']), float(g['tp'])
    assert (W, B * P, V) == (256, 4096, 8)
    m = IDRNetwork(ConfigDict(synth.model_conf(W)))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, seed).items()})
    m = m.cuda().train().set_trace_dtype('bf16')
    inp, gt = synth.make_batch(B, P, V, seed=seed, size=float(g['scene_size']), center=tuple(g['scene_center']),
                               feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    torch.manual_seed(seed + 5)
    out = m({k: t(v) for k, v in inp.items()}, tp)
    mask, mref = out['network_object_mask'].cpu().numpy(), g['out_network_object_mask']
    agree = float((mask == mref).mean())
    both = mask & mref
    cam = np.repeat(inp['pose'][:, :3, 3], P, axis=0)
    depth = np.linalg.norm(out['points'].detach().cpu().numpy() - cam, axis=1)
    dref = np.linalg.norm(g['out_points'] - cam, axis=1)
    rel = np.abs(depth - dref)[both] / dref[both]
    lo = IDRLoss()(out, {k: t(v) for k, v in gt.items()}, tp, B)
    print('c5 share, bf16 tracer vs fp32 reference: masks agree %.4f, depth rel p99 %.3g max %.3g' % (agree, np.percentile(rel, 99), rel.max()))
    # the budget at its MEASURED values (round 3: masks 99.95 % = 3 flips of 4096, p99 1.5e-3, max 8.5e-3): a regression that triples the flips fails
    assert agree >= 0.999 and np.percentile(rel, 99) < 2e-3
    n_bad = int((rel > 1e-2).sum())
    print('   tail: rays with depth error > 1e-2: %d of %d' % (n_bad, rel.size))
    assert rel.max() < 1e-2 and n_bad == 0
    for k in ('loss', 'rgb_loss', 'eikonal_loss', 'depth_loss', 'feat_loss', 'surf_loss'):
        v, ref = float(lo[k].detach().reshape(-1)[0]), float(g['loss_' + k])
        print('   %s %.6g (reference %.6g)' % (k, v, ref))
        assert abs(v - ref) <= 0.03 * max(abs(ref), 1e-3), (k, v, ref)
    lo['loss'].backward()
    assert torch.isfinite(torch.cat([p.grad.flatten() for p in m.parameters()])).all()


# ---- weights-only bf16 (trace_dtype 2, IDRNetwork.set_trace_dtype('bf16w')): bf16-rounded weights, fp32 activations on the fp32 MFMA.
# BASELINE configs[4] asks for "bf16 MLP weights"; this mode is exactly that and nothing more, and it is BIT-CHECKABLE: the fp32 engine on
# rounded weights equals the oracle on rounded weights.  Against it the full bf16 engine's extra error (activation rounding) is a number.
@pytest.mark.parametrize('W', [64, 256, 512])
def test_weights_only_bf16_mlp_bit_exact_vs_oracle(oracle, W):
    sd = synth.make_state_dict(W, 0)
    net = ops.pack_bf16_net(sdf_packed_net(sd), weights_only=True)
    rs = np.random.RandomState(3)
    x = rs.uniform(-1.2, 1.2, size=(4000, 3)).astype(np.float32)
    ref = oracle.sdf_forward(oracle.Net(sd, bf16='weights'), x, ncols=1)[:, 0]
    f32 = oracle.sdf_forward(oracle.Net(sd), x, ncols=1)[:, 0]
    for mt in (1, 2, 4):
        y = ops.sdf_col0(net, t(x), mt=mt).cpu().numpy()
        assert np.array_equal(y, ref), 'mt=%d max diff %g' % (mt, np.abs(y - ref).max())
    print('W=%d weights-only bf16 vs fp32 network: max %.3g mean %.3g' % (W, np.abs(ref - f32).max(), np.abs(ref - f32).mean()))
    assert not np.array_equal(ref, f32)


@pytest.mark.parametrize('W,mode', [(64, 'train'), (256, 'eval'), (256, 'train')])
def test_weights_only_bf16_tracer_bit_exact_vs_oracle_and_budget(oracle, W, mode):
    g = golden('trace_mlp_w%d_%s' % (W, mode))
    sd = synth.make_state_dict(W, int(g['seed']))
    net = ops.pack_bf16_net(sdf_packed_net(sd), weights_only=True)
    B, P = int(g['B']), int(g['P'])
    cam, dirs = t(g['cam_loc']), t(g['ray_dirs']).reshape(B, P, 3)
    om = torch.ones(B * P, dtype=torch.bool, device='cuda')
    training = mode == 'train'
    iv = torch.linspace(0, 1, 100)
    pts, mask, dists, cnt = ops.trace(net, cam, dirs, om, trace_params(W), training, iv.cuda(), t(g['minsdf_steps']), mt=1, mt_samples=2)
    p_o, m_o, d_o, rows = oracle.trace(oracle.Net(sd, bf16='weights'), g['cam_loc'], g['ray_dirs'], np.ones(B * P, bool), training, g['minsdf_steps'],
                                       iv.numpy(), **synth.model_conf(W)['ray_tracer'])
    mask, dists = mask.cpu().numpy(), dists.cpu().numpy()
    assert np.array_equal(mask, m_o) and np.array_equal(dists, d_o) and np.array_equal(pts.cpu().numpy(), p_o)     # bit for bit
    assert np.array_equal(cnt.cpu().numpy()[:4], rows)
    agree_r = (mask == g['mask']).mean()
    both = mask & g['mask']
    rel_r = np.abs(dists - g['dists'])[both] / np.abs(g['dists'][both])
    print('W=%d %s weights-only bf16 vs fp32 reference: masks agree %.4f, hit depth rel max %.3g 99%% %.3g median %.3g, rays > 1e-2: %d' % (
        W, mode, agree_r, rel_r.max(), np.percentile(rel_r, 99), np.median(rel_r), int((rel_r > 1e-2).sum())))
    assert agree_r >= 0.98 and np.percentile(rel_r, 99) < 5e-3


def test_what_rounding_the_activations_costs_at_the_c5_share():
    """idr_c5share (fp32 reference, 4096 rays, V = 8): the step with the weights-only-bf16 tracer and with the full bf16 engine side by side --
    mask agreement, depth error percentiles / maximum, loss deviation.  The difference between the two columns is the price of rounding
    the activations (and of the hardware exp / log softplus) on top of what BASELINE's "bf16 MLP weights" asks for."""
    from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
    from mvsdf_amd.model.loss import IDRLoss
    from mvsdf_amd.utils.config import ConfigDict
    g = golden('idr_c5share')
    W, B, P, V, seed, tp = int(g['W']), int(g['B']), int(g['P']), int(g['V']), int(g['seed']), float(g['tp'])
    inp, gt = synth.make_batch(B, P, V, seed=seed, size=float(g['scene_size']), center=tuple(g['scene_center']),
                               feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    cam = np.repeat(inp['pose'][:, :3, 3], P, axis=0)
    dref = np.linalg.norm(g['out_points'] - cam, axis=1)
    mref = g['out_network_object_mask']
    res = {}
    for dt in ('f32', 'bf16w', 'bf16'):
        m = IDRNetwork(ConfigDict(synth.model_conf(W)))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, seed).items()})
        m = m.cuda().train().set_trace_dtype(dt)
        torch.manual_seed(seed + 5)
        out = m({k: t(v) for k, v in inp.items()}, tp)
        mask = out['network_object_mask'].cpu().numpy()
        both = mask & mref
        depth = np.linalg.norm(out['points'].detach().cpu().numpy() - cam, axis=1)
        rel = np.abs(depth - dref)[both] / dref[both]
        lo = IDRLoss()(out, {k: t(v) for k, v in gt.items()}, tp, B)
        dl = max(abs(float(lo[k].detach().reshape(-1)[0]) - float(g['loss_' + k])) / max(abs(float(g['loss_' + k])), 1e-3)
                 for k in ('loss', 'rgb_loss', 'eikonal_loss', 'depth_loss', 'feat_loss', 'surf_loss'))
        res[dt] = (float((mask == mref).mean()), float(np.median(rel)), float(np.percentile(rel, 99)), float(rel.max()), int((rel > 1e-2).sum()), dl)
        print('c5 share, tracer %-5s vs fp32 reference: masks agree %.4f, depth rel median %.3g p99 %.3g max %.3g, rays > 1e-2: %d, worst loss term off by %.3g' % ((dt,) + res[dt]))
    assert res['f32'][0] == 1.0 and res['f32'][3] < 1e-4
    assert res['bf16w'][0] >= 0.995 and res['bf16w'][2] < 2e-3
    assert res['bf16w'][2] <= res['bf16'][2] * 1.5 + 1e-4          # the weights-only mode is not (much) worse than the full bf16 engine anywhere
