"""trace_dtype 'f32x3': the reference's fp32 tracing arithmetic (idr.py:77-94 inside ray_tracing.py:27-98, fp32 weights UNROUNDED) on gfx950's bf16
matrix cores -- weights and activations as three bf16 terms each (w = w0 + w1 + w2, a = a0 + a1 + a2 exactly), the six exact products a_s w_j with
s + j <= 2, fp32 accumulators (csrc/tile_engine_bf16s.h::mv_gemm_rolling_bw).

Two oracles:
  * `oracle.Net(sd, bf16='f32x3')` restates THIS arithmetic, matrix instruction included (oracle_mvsdf.c::sdf_row_f32x3, the model of
    v_mfma_f32_16x16x32_bf16 from tools/micro/mfma_bf16_model/): the engine equals it BIT FOR BIT -- MLP values and every tracer output
    (masks, dists, points, row counters), like the 'f32' engine equals the fmaf-chain oracle; that oracle is pinned to the reference's fixtures on CPU
    (tests/test_oracle_golden.py);
  * against the fmaf-chain oracle (another summation order) and the reference's own fixtures: SDF values no further from an fp64 evaluation than
    the fp32 chain is; hit masks identical except rays whose recorded decision margin is below 1e-6 (none so far), hit depths within 1e-4; every
    end-to-end reference fixture of tests/test_gpu_idr.py passes with this tracer unchanged."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from helpers import sdf_packed_net, t, trace_params
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
from test_gpu_bf16s import _compare_with_oracle

pytestmark = pytest.mark.gpu


def _net(sd):
    return ops.pack_bf16_net(sdf_packed_net(sd), terms=3, weight_terms=3)


def _f64_sdf(onet, x):
    """the oracle network's folded fp32 weights evaluated in fp64 with exact elementary functions (oracle/oracle_np.py)"""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle_np

    class N:
        pass
    m = N()
    m.W = [w.astype(np.float64) for w in onet.W]
    m.b = [b.astype(np.float64) for b in onet.b]
    m.n_layers, m.multires, m.skip_layers = onet.n_layers, onet.multires, onet.skip_in
    return oracle_np.sdf_forward(m, x, need_normal=False)[0][:, 0]


@pytest.mark.parametrize('W,n', [(64, 20000), (256, 2500), (512, 600)])
def test_three_term_mlp_bit_exact_vs_its_oracle(oracle, W, n):
    """every row tiling and both weight-fetch schemes against the oracle's model of the matrix instruction"""
    for seed in (0, 7):
        sd = synth.make_state_dict(W, seed)
        x = np.random.RandomState(3 + seed).uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
        x[:8] *= 1e-3                                                                  # near the origin: sin(2^m x) ~ x, tiny positional-encoding terms
        x[8:16] = 0.0
        x[16:24] = np.float32(1e-30)                                                   # below the engine's 2^-40 flush
        ref = oracle.sdf_forward(oracle.Net(sd, bf16='f32x3'), x, ncols=1)[:, 0]
        net = _net(sd)
        for mt in (1, 2, 4, 49):
            if mt == 49 and W > 256:
                continue
            y = ops.sdf_col0(net, t(x), mt=mt).cpu().numpy()
            bad = np.nonzero(y.view(np.uint32) != ref.view(np.uint32))[0]
            assert bad.size == 0, (W, seed, mt, bad[:5], y[bad[:5]], ref[bad[:5]])


@pytest.mark.parametrize('W,mode,views,rays', [(64, 'train', 4, 1024), (64, 'eval', 4, 1024), (256, 'train', 4, 256), (256, 'eval', 4, 256), (512, 'train', 2, 192), (512, 'eval', 1, 128),
                                               (64, 'eval_render', 4, 1024), (64, 'train_render', 4, 512), (256, 'eval_render', 2, 256)])
def test_three_term_tracer_bit_exact_vs_its_oracle(oracle, W, mode, views, rays):
    """RayTracing.forward on the trace_mlp fixtures' rays (all 1024 at W = 256, 384 + 128 at W = 512: the AVX2 form of the instruction model costs the CPU
    2.6 / ~12 ms per MLP row and thread): masks, dists, points and the row counters equal the oracle's bit for bit, for every chunking"""
    g = golden('trace_mlp_w%d_%s' % (W, mode))
    sd = synth.make_state_dict(W, int(g['seed']))
    B, P = int(g['B']), int(g['P'])
    dirs = np.ascontiguousarray(g['ray_dirs'].reshape(B, P, 3)[:views, :rays])
    cam = np.ascontiguousarray(g['cam_loc'][:views])
    om = np.ones(views * rays, bool)
    iv = torch.linspace(0, 1, 100)
    # (*_render: the IDR_RENDER variant -- dist_clip 0.05, 40 iterations, ray_tracing.py:127-131 -- with the same two numbers on both sides)
    from helpers import render_overrides
    over = render_overrides(g)
    assert bool(over) == ('render' in mode)
    training = mode.startswith('train')
    p_o, m_o, d_o, rows = oracle.trace(oracle.Net(sd, bf16='f32x3'), cam, dirs, om, training, g['minsdf_steps'], iv.numpy(), **dict(synth.model_conf(W)['ray_tracer'], **over))
    ref_mask = g['mask'].reshape(B, P)[:views, :rays].reshape(-1)
    ref_d = g['dists'].reshape(B, P)[:views, :rays].reshape(-1)
    assert np.array_equal(m_o, ref_mask)                                           # the oracle of this arithmetic against the reference's own masks on these rays
    rel = np.abs(d_o - ref_d)[ref_mask] / np.abs(ref_d[ref_mask])
    assert np.sort(rel)[-3 if 'render' in mode else -1] < 1e-4, np.sort(rel)[-4:]  # (render: up to two recorded ties, helpers.depth_check)
    net = _net(sd)
    for mt, mts in ((1, 2), (2, 4), (4, 1)):
        pts, mask, dists, cnt = ops.trace(net, t(cam), t(dirs), t(om), trace_params(W, **over), training, iv.cuda(), t(g['minsdf_steps']), mt=mt, mt_samples=mts)
        assert np.array_equal(mask.cpu().numpy(), m_o), (mt, mts)
        assert np.array_equal(dists.cpu().numpy(), d_o), (mt, mts)
        assert np.array_equal(pts.cpu().numpy(), p_o), (mt, mts)
        c = cnt.cpu().numpy()
        assert np.array_equal(c[:4], rows) and c[8] <= c[1]


@pytest.mark.parametrize('W', [64, 256, 512])
def test_three_term_mlp_is_as_close_to_fp64_as_the_fp32_chain(oracle, W):
    sd = synth.make_state_dict(W, 0)
    x = np.random.RandomState(3).uniform(-1.2, 1.2, size=(20000, 3)).astype(np.float32)
    onet = oracle.Net(sd)
    ref = _f64_sdf(onet, x)
    chain = oracle.sdf_forward(onet, x, ncols=1)[:, 0].astype(np.float64)
    net = _net(sd)
    ys = [ops.sdf_col0(net, t(x), mt=mt).cpu().numpy() for mt in (1, 2, 4)]
    assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])          # row tiling does not change a row's arithmetic
    y = ys[0].astype(np.float64)
    rms = lambda d: float(np.sqrt((d ** 2).mean()))
    e_chain, e_split = chain - ref, y - ref
    print('W=%d: |fp32 chain - fp64| max %.3g rms %.3g;  |f32x3 - fp64| max %.3g rms %.3g;  |f32x3 - chain| max %.3g' % (
        W, np.abs(e_chain).max(), rms(e_chain), np.abs(e_split).max(), rms(e_split), np.abs(y - chain).max()))
    assert not np.array_equal(y, chain)                                            # (another arithmetic: a bit-equal result would mean the fp32 engine ran)
    assert rms(e_split) <= 1.1 * rms(e_chain) and np.abs(e_split).max() <= 1.5 * np.abs(e_chain).max()
    assert np.abs(y - chain).max() < 5e-6


@pytest.mark.parametrize('W,mode', [(64, 'train'), (256, 'eval'), (256, 'train'), (512, 'train')])
def test_three_term_tracer_vs_the_fp32_oracle(oracle, W, mode):
    g = golden('trace_mlp_w%d_%s' % (W, mode))
    sd = synth.make_state_dict(W, int(g['seed']))
    B, P = int(g['B']), int(g['P'])
    dirs = g['ray_dirs'].reshape(B, P, 3)
    mask, dists, ndiff = _compare_with_oracle('trace_mlp_w%d_%s f32' % (W, mode), oracle, sd, W, g['cam_loc'], dirs, np.ones(B * P, bool), mode == 'train',
                                              g['minsdf_steps'], 3, 1, 2, net=_net(sd), onet=oracle.Net(sd))
    iv = torch.linspace(0, 1, 100)
    p2, m2, d2, _ = ops.trace(_net(sd), t(g['cam_loc']), t(dirs), torch.ones(B * P, dtype=torch.bool, device='cuda'), trace_params(W), mode == 'train', iv.cuda(),
                              t(g['minsdf_steps']), mt=2, mt_samples=4)
    assert np.array_equal(m2.cpu().numpy(), mask) and np.array_equal(d2.cpu().numpy(), dists)      # all chunkings give the same result
    # and against the imported PyTorch reference's own outputs on these rays (the fixture): the same bar as the bit-exact engine's test
    both = mask & g['mask']
    rel = np.abs(dists - g['dists'])[both] / np.abs(g['dists'][both])
    print('   vs the reference fixture: masks differ on %d rays, hit depth rel max %.3g' % (int((mask != g['mask']).sum()), rel.max()))
    assert int((mask != g['mask']).sum()) <= ndiff + 1 and np.percentile(rel, 99.9) < 1e-4


@pytest.mark.parametrize('name', ['idr_c2', 'idr_c3', 'idr_c5share'])
def test_three_term_tracer_at_the_baseline_shares_vs_oracle(oracle, name):
    """the ray batches of BASELINE configs[1], [2] and one GPU's share of [4] (training mode, the object mask of the batch)"""
    g = golden(name)
    W, B, P, V, seed = int(g['W']), int(g['B']), int(g['P']), int(g['V']), int(g['seed'])
    sd = synth.make_state_dict(W, seed)
    inp, _ = synth.make_batch(B, P, V, seed=seed, size=float(g['scene_size']), center=tuple(g['scene_center']),
                              feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    dirs, cam = ops.camera_rays(t(inp['uv']), t(inp['pose']), t(inp['intrinsics']))
    steps = np.random.RandomState(seed).uniform(size=100).astype(np.float32)
    om = np.asarray(inp['object_mask']).reshape(-1).astype(bool)
    _compare_with_oracle(name, oracle, sd, W, cam.cpu().numpy(), dirs.cpu().numpy(), om, True, steps, 3, 2, 2, net=_net(sd), onet=oracle.Net(sd))


@pytest.mark.parametrize('name', ['idr_c2', 'idr_c5share'])
def test_three_term_tracer_bit_exact_on_the_full_baseline_batches(oracle, name):
    """ALL 2048 rays of BASELINE configs[1] (`idr_c2`: 188 k MLP rows) and all 4096 rays of one GPU's share of configs[4] (`idr_c5share`: 373 k rows), training
    mode with the batch's object mask, against the instruction-model oracle: masks, dists, points, row counters bit for bit -- the check that used to be a
    one-off (tools/micro/f32s/bitexact_c2.py, 208 s / 412 s) runs in the suite with the eight-column form of the model."""
    import time
    g = golden(name)
    W, B, P, V, seed = int(g['W']), int(g['B']), int(g['P']), int(g['V']), int(g['seed'])
    sd = synth.make_state_dict(W, seed)
    inp, _ = synth.make_batch(B, P, V, seed=seed, size=float(g['scene_size']), center=tuple(g['scene_center']),
                              feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    dirs, cam = ops.camera_rays(t(inp['uv']), t(inp['pose']), t(inp['intrinsics']))
    steps = np.random.RandomState(seed).uniform(size=100).astype(np.float32)
    om = np.asarray(inp['object_mask']).reshape(-1).astype(bool)
    iv = torch.linspace(0, 1, 100)
    pts, mask, dists, cnt = ops.trace(_net(sd), cam, dirs, t(om), trace_params(W), True, iv.cuda(), t(steps), mt=1, mt_samples=4)
    t0 = time.time()
    p_o, m_o, d_o, rows = oracle.trace(oracle.Net(sd, bf16='f32x3'), cam.cpu().numpy(), dirs.cpu().numpy(), om, True, steps, iv.numpy(), **synth.model_conf(W)['ray_tracer'])
    print('%s: %d rays, %d oracle MLP rows in %.0f s on %d threads; hits %d' % (name, B * P, int(rows.sum()), time.time() - t0, oracle.lib().orc_num_threads(), int(m_o.sum())))
    assert np.array_equal(mask.cpu().numpy(), m_o) and np.array_equal(dists.cpu().numpy(), d_o) and np.array_equal(pts.cpu().numpy(), p_o)
    assert np.array_equal(cnt.cpu().numpy()[:4], rows)
    # ... and the reference's own hit masks on these rays (the imported PyTorch reference's output in the fixture)
    assert np.array_equal(m_o, g['out_network_object_mask'].reshape(-1).astype(bool))


def test_reference_fixtures_pass_with_the_fmaf_chain_tracer():
    """Every end-to-end fixture of the imported reference (tests/test_gpu_idr.py: c1, c2, c3, the c5 share, W = 512, phase 0, the 4-step replay) ran on the
    product default 'f32x3' in the main suite; here once more on the fmaf-chain arithmetic `set_trace_dtype('f32')` (MVSDF_TEST_TRACE_DTYPE: tests/conftest.py) --
    same assertions, same tolerances."""
    e = dict(os.environ, MVSDF_TEST_TRACE_DTYPE='f32')
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider', 'tests/test_gpu_idr.py'], cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    tail = p.stdout.decode(errors='replace')[-2500:]
    assert p.returncode == 0, tail
    assert ' passed' in tail and 'failed' not in tail, tail


def test_step_driver_lazy_outputs_options_and_data_parallel_suites_pass_with_the_fmaf_chain_tracer():
    """The suites that exercise the tracer through the rest of the product (native step driver vs the Python route, deferred outputs, constructor
    options, eval rendering / mesh grid, 8 ranks on one GPU) once more with the fmaf-chain arithmetic 'f32' (the main suite runs them on 'f32x3')."""
    e = dict(os.environ, MVSDF_TEST_TRACE_DTYPE='f32')
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '-p', 'no:cacheprovider', 'tests/test_gpu_shapes.py', 'tests/test_gpu_native_step.py',
                        'tests/test_gpu_lazy.py', 'tests/test_gpu_options.py', 'tests/test_gpu_dp.py'], cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    tail = p.stdout.decode(errors='replace')[-2500:]
    assert p.returncode == 0, tail
    assert ' passed' in tail and 'failed' not in tail, tail


def test_three_term_step_vs_the_bit_exact_step_at_c2():
    """The whole training step of BASELINE configs[1] (2048 rays, 4 source views) with the f32x3 tracer against the same step with the bit-exact tracer."""
    from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
    from mvsdf_amd.model.loss import IDRLoss
    from mvsdf_amd.utils.config import ConfigDict
    g = golden('idr_c2')
    W, B, P, V, seed, tp = int(g['W']), int(g['B']), int(g['P']), int(g['V']), int(g['seed']), float(g['tp'])
    inp, gt = synth.make_batch(B, P, V, seed=seed, size=float(g['scene_size']), center=tuple(g['scene_center']),
                               feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    cam = np.repeat(inp['pose'][:, :3, 3], P, axis=0)
    res = {}
    for dt in ('f32', 'f32x3'):
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
    (m0, d0, l0, g0), (m1, d1, l1, g1) = res['f32'], res['f32x3']
    ndiff = int((m0 != m1).sum())
    both = m0 & m1
    rel = np.abs(d1 - d0)[both] / d0[both]
    dl = max(abs(l1[k] - l0[k]) / max(abs(l0[k]), 1e-3) for k in l0)
    print('c2 step, f32x3 vs f32: masks differ on %d rays, hit depth rel max %.3g p99 %.3g, %d rays beyond 1e-4, worst loss term off by %.3g, |grad| %.6g vs %.6g' % (
        ndiff, rel.max(), np.percentile(rel, 99), int((rel > 1e-4).sum()), dl, g1, g0))
    assert ndiff <= 1 and int((rel > 1e-4).sum()) <= 1 and np.percentile(rel, 99) < 5e-6
    assert dl <= 2e-4 and abs(g1 - g0) <= 1e-3 * g0


@pytest.mark.parametrize('seed', [11, 12, 13, 14, 15, 16])
def test_three_term_tracer_fuzz_vs_oracle(oracle, seed):
    """Randomised scenes / cameras / masks / tracer parameters (the draws of tests/test_gpu_trace.py::test_trace_fuzz_bit_exact_vs_oracle, W = 64 and 256):
    hit masks equal to the fp32 oracle's except rays whose recorded decision margin is below 1e-6, hit depths within 1e-4, identical results
    for every chunking."""
    rs = np.random.RandomState(seed)
    W = 64 if seed % 2 else 256
    sd = synth.make_state_dict(W, seed)
    onet, net = oracle.Net(sd), _net(sd)
    B, P = int(rs.randint(1, 4)), int(rs.randint(40, 300))
    inp, _ = synth.make_batch(B, P, 0, seed=seed, radius=float(rs.uniform(1.6, 3.0)), height=float(rs.uniform(-0.5, 1.2)),
                              focal_scale=float(rs.uniform(0.8, 2.5)), with_features=False)
    dirs, cam = oracle.camera_rays(inp['uv'], inp['pose'], inp['intrinsics'])
    om = rs.uniform(size=B * P) < rs.choice([1.1, 0.6])
    tr = dict(synth.model_conf(W)['ray_tracer'])
    tr['n_steps'] = int(rs.choice([100, 37, 16, 9]))
    tr['sphere_tracing_iters'] = int(rs.choice([10, 3, 1]))
    tr['n_secant_steps'] = int(rs.choice([8, 3]))
    tr['line_step_iters'] = int(rs.choice([3, 1, 0]))
    iv = torch.linspace(0, 1, steps=tr['n_steps']).numpy()
    steps = rs.uniform(size=tr['n_steps']).astype(np.float32)
    params = (tr['object_bounding_sphere'], tr['sdf_threshold'], tr['line_search_step'], tr['line_step_iters'], tr['sphere_tracing_iters'],
              tr['n_steps'], tr['n_secant_steps'], 0.5)
    for training in (True, False):
        p_o, m_o, d_o, rows_o, mg = oracle.trace(onet, cam, dirs, om, training, steps, iv, margins=True, **tr)
        margin = mg.min(axis=1)
        first = None
        if W == 64:                                      # ... and against this arithmetic's own oracle: everything bit for bit
            p3, m3, d3, rows3 = oracle.trace(oracle.Net(sd, bf16='f32x3'), cam, dirs, om, training, steps, iv, **tr)
            pts, mask, dists, cnt = ops.trace(net, t(cam), t(dirs), t(om), params, training, t(iv), t(steps), mt=1, mt_samples=2)
            assert np.array_equal(mask.cpu().numpy(), m3) and np.array_equal(dists.cpu().numpy(), d3) and np.array_equal(pts.cpu().numpy(), p3)
            assert np.array_equal(cnt.cpu().numpy()[:4], rows3)
        for mt, mts in ((1, 1), (2, 2), (4, 4)):
            pts, mask, dists, cnt = ops.trace(net, t(cam), t(dirs), t(om), params, training, t(iv), t(steps), mt=mt, mt_samples=mts)
            mask, dists = mask.cpu().numpy(), dists.cpu().numpy()
            if first is None:
                first = (mask, dists)
                diff = np.nonzero(mask != m_o)[0]
                both = mask & m_o
                rel = np.abs(dists - d_o) / np.maximum(np.abs(d_o), 1e-12)
                bad = np.nonzero(both & (rel > 1e-4))[0]
                print('seed %d W=%d %s (%d rays, n_steps %d, iters %d): masks differ on %d, rays beyond 1e-4: %d, hit depth rel max %.3g' % (
                    seed, W, 'train' if training else 'eval', B * P, tr['n_steps'], tr['sphere_tracing_iters'], diff.size, bad.size, rel[both].max() if both.any() else 0.0))
                assert all(margin[i] < 1e-6 for i in diff), [(int(i), float(margin[i])) for i in diff]
                assert all(margin[i] < 1e-6 for i in bad), [(int(i), float(rel[i]), float(margin[i])) for i in bad]
            else:
                assert np.array_equal(mask, first[0]) and np.array_equal(dists, first[1]), (training, mt)
