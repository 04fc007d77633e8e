#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz by importing the PyTorch
reference (jzhangbs/MVSDF @ /root/reference) on CPU.

Runs ONLY in the build container (the reference never travels to the GPU box).
The fixtures hold inputs' seeds/recipes and the reference's OUTPUTS; weights are
regenerated from mvsdf_amd.utils.synth (checksummed), never committed.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Shim: stub imageio/skimage/cv2 + numpy.lib.function_base,
Tensor.cuda -> identity, dict-backed conf object standing in for pyhocon.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from mvsdf_amd.utils import synth  # noqa: E402

for n in ('imageio', 'skimage', 'cv2'):
    sys.modules[n] = types.ModuleType(n)
fb = types.ModuleType('numpy.lib.function_base')
fb.diff = np.diff
sys.modules['numpy.lib.function_base'] = fb
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
sys.path.insert(0, '/root/reference/code')


class Conf(dict):
    def _g(s, k):
        for p in k.split('.'):
            s = s[p]
        return s
    get_int = lambda s, k: int(s._g(k))
    get_float = lambda s, k: float(s._g(k))
    get_config = lambda s, k: Conf(s._g(k))


from model.implicit_differentiable_renderer import IDRNetwork  # noqa: E402
from model.loss import IDRLoss  # noqa: E402
from model.ray_tracing import RayTracing  # noqa: E402
from model.sample_network import SampleNetwork  # noqa: E402
from utils import rend_util  # noqa: E402

torch.set_default_dtype(torch.float32)
torch.set_num_threads(1)   # reference setting (idr_train.py:21)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def build_model(W, seed, skip_in=(4,), **kw):
    with quiet():
        m = IDRNetwork(Conf(synth.model_conf(W, skip_in=skip_in, **kw)))
    sd = synth.make_state_dict(W, seed, skip_in=skip_in)
    m.load_state_dict({k: T(v) for k, v in sd.items()})
    return m, sd


OUT_DIR = HERE                                                  # --check regenerates into a scratch directory


def save(name, **arrs):
    path = os.path.join(OUT_DIR, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


def analytic_sdf(x):
    """Tier-0 SDF (SURVEY.md section 4 idea, polynomial bumps instead of sin/cos): built from
    +,-,* only (torch's CPU sqrt is not correctly rounded), so the C oracle (oracle_mvsdf.c::analytic_sdf) reproduces it bit for bit.
    Not 1-Lipschitz on purpose: exercises the line search, the sampler and the secant."""
    X, Y, Z = x[:, 0], x[:, 1], x[:, 2]
    x2, y2, z2 = X * X, Y * Y, Z * Z
    r2 = x2 + y2 + z2
    t5 = ((16.0 * x2 - 20.0) * x2 + 5.0) * X
    t4 = (8.0 * y2 - 8.0) * y2 + 1.0
    t3 = (4.0 * z2 - 3.0) * Z
    return 1.4 * (r2 - 0.36) + 0.2 * (t5 * t4 * t3)


# ----------------------------------------------------------------------------------------------
def g_sdf(W, n, seed):
    m, sd = build_model(W, seed)
    rs = np.random.RandomState(seed + 7)
    x = rs.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    x[: n // 8] *= 1.6           # some points outside the unit cube
    net = m.implicit_network
    net.eval()
    with torch.no_grad():
        out = net(T(x)).numpy()
    xg = T(x).clone()
    g = net.gradient(xg)[:, 0, :].detach().numpy()
    # folded weights of layer 0/8 as torch computes them (pins the weight-norm fold)
    w0 = net.lin0.weight.detach().numpy()
    w8 = net.lin8.weight.detach().numpy()
    save('sdf_w%d' % W, W=W, seed=seed, x=x, out=out, grad=g, w0=w0, w8_row0=w8[0],
         checksum=synth.state_checksum(sd))


def g_render(W, n, seed):
    m, sd = build_model(W, seed)
    rs = np.random.RandomState(seed + 11)
    pts = rs.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    nrm = rs.normal(size=(n, 3)).astype(np.float32)
    view = rs.normal(size=(n, 3)).astype(np.float32)
    view /= np.linalg.norm(view, axis=1, keepdims=True)
    feat = rs.normal(size=(n, synth.FEAT)).astype(np.float32)
    with torch.no_grad():
        rgb = m.rendering_network(T(pts), T(nrm), T(view), T(feat)).numpy()
    save('render_w%d' % W, W=W, seed=seed, points=pts, normals=nrm, view=view, feat=feat, rgb=rgb,
         checksum=synth.state_checksum(sd))


def g_rays(seed):
    inp, _ = synth.make_batch(3, 700, 0, seed=seed, focal_scale=0.9, with_features=False)
    # add a skewed intrinsic to exercise lift()'s skew terms (rend_util.py:96-97)
    inp['intrinsics'][1, 0, 1] = 3.5
    with torch.no_grad():
        dirs, cam_loc = rend_util.get_camera_params(T(inp['uv']), T(inp['pose']), T(inp['intrinsics']))
        si, mi = rend_util.get_sphere_intersection(cam_loc, dirs, r=1.0)
    save('rays', seed=seed, uv=inp['uv'], pose=inp['pose'], intrinsics=inp['intrinsics'],
         ray_dirs=dirs.numpy(), cam_loc=cam_loc.numpy(), sphere_intersections=si.numpy(),
         mask_intersect=mi.numpy())


class CountingSDF:
    def __init__(self, f):
        self.f, self.rows = f, []

    def __call__(self, x):
        self.rows.append(x.shape[0])
        return self.f(x)


class MarginRecorder:
    """Per-ray decision margins of one RayTracing.forward call of the reference, so that a mask or depth
    difference in a parity test can be attributed to a near-tie instead of to a bug.  Wraps the `sdf` callable: every evaluated row
    x = cam + t * dir is matched to its ray (float64, smallest angle to a ray of any view) and the values the REFERENCE's network
    returned are reduced per ray to
      min_abs_sdf : min |sdf|            -- distance of any sign test from flipping (line search ray_tracing.py:175,192, sampler :230,
                                            secant :266-272);
      min_thr_gap : min |sdf - threshold| -- distance of any convergence test from flipping (ray_tracing.py:149-157);
      n_evals     : evaluations of that ray;
    and wraps `sphere_tracing` to keep acc_gap = acc_end_dis - acc_start_dis at its return, the quantity whose sign is the initial hit
    mask (ray_tracing.py:41) and the sampler's range (ray_tracing.py:48-49)."""

    def __init__(self, rt, cam_loc, dirs):
        self.cam = cam_loc.detach().double()                     # [B,3]
        self.dirs = dirs.detach().double()                       # [B,P,3]
        self.dirs = self.dirs / self.dirs.norm(dim=-1, keepdim=True)    # unit in float64 (the float32 directions are unit to 6e-8 only)
        B, P, _ = self.dirs.shape
        self.R = B * P
        self.min_abs = np.full(self.R, np.inf)
        self.min_thr = np.full(self.R, np.inf)
        self.n_evals = np.zeros(self.R, np.int64)
        self.acc_gap = None
        self.thr = float(rt.sdf_threshold)
        self.worst_match = 0.0
        inner = rt.sphere_tracing

        def sphere_tracing(*a, **k):
            res = inner(*a, **k)
            self.acc_gap = (res[3] - res[2]).detach().numpy().copy()
            return res
        rt.sphere_tracing = sphere_tracing

    def wrap(self, sdf):
        def f(x):
            y = sdf(x)
            self.record(x.detach(), y.detach().reshape(-1))
            return y
        return f

    def record(self, x, y):
        B, P, _ = self.dirs.shape
        for lo in range(0, x.shape[0], 16384):
            xc = x[lo:lo + 16384].double()
            best = torch.full((xc.shape[0],), -2.0, dtype=torch.float64)
            ray = torch.zeros(xc.shape[0], dtype=torch.int64)
            for b in range(B):
                v = xc - self.cam[b]
                u = v / v.norm(dim=1, keepdim=True)
                c, i = (u @ self.dirs[b].T).max(1)
                upd = c > best
                best[upd] = c[upd]
                ray[upd] = b * P + i[upd]
            self.worst_match = max(self.worst_match, float((1 - best).max()))
            yc = y[lo:lo + 16384].double().numpy()
            r = ray.numpy()
            np.minimum.at(self.min_abs, r, np.abs(yc))
            np.minimum.at(self.min_thr, r, np.abs(yc - self.thr))
            np.add.at(self.n_evals, r, 1)

    def arrays(self):
        assert self.worst_match < 1e-10, self.worst_match          # 1 - cos of the angle between a row and its ray: every row sits on a ray
        return dict(margin_min_abs_sdf=self.min_abs.astype(np.float32), margin_min_thr_gap=self.min_thr.astype(np.float32),
                    margin_n_evals=self.n_evals, margin_acc_gap=self.acc_gap.astype(np.float32))


def run_tracer(sdf, cam_loc, object_mask, dirs, training, seed, margins=False, **tr):
    rt = RayTracing(**tr)
    rt.train(training)
    rec = None
    if margins:
        rec = MarginRecorder(rt, cam_loc, dirs)
        sdf = rec.wrap(sdf)
    torch.manual_seed(seed)
    steps = torch.empty(100).uniform_(0.0, 1.0).numpy()     # what minimal_sdf_points will draw first
    torch.manual_seed(seed)
    c = CountingSDF(sdf)
    with torch.no_grad(), quiet():
        pts, mask, dists = rt(sdf=c, cam_loc=cam_loc, object_mask=object_mask, ray_directions=dirs)
    with torch.no_grad():
        si, mi = rend_util.get_sphere_intersection(cam_loc, dirs, r=tr.get('object_bounding_sphere', 1.0))
    extra = dict(intervals=torch.linspace(0, 1, steps=tr.get('n_steps', 100)).numpy(),
                 sphere_intersections=si.numpy(), mask_intersect=mi.numpy())
    if rec is not None:
        extra.update(rec.arrays())
    return pts.numpy(), mask.numpy(), dists.numpy(), steps, np.array(c.rows, dtype=np.int64), extra


def g_trace_analytic(seed):
    tr = synth.model_conf(64)['ray_tracer']
    inp, _ = synth.make_batch(4, 3000, 0, seed=seed, focal_scale=0.9, with_features=False)
    with torch.no_grad():
        dirs, cam_loc = rend_util.get_camera_params(T(inp['uv']), T(inp['pose']), T(inp['intrinsics']))
    rs = np.random.RandomState(seed + 3)
    omask = rs.uniform(size=(4 * 3000,)) < 0.8                # exercise object_mask paths (use_mask=True style)
    for training in (False, True):
        for mname, om in (('ones', np.ones_like(omask)), ('rand', omask)):
            pts, mask, dists, steps, rows, extra = run_tracer(analytic_sdf, cam_loc, T(om), dirs, training, seed + 5, **tr)
            save('trace_analytic_%s_%s' % ('train' if training else 'eval', mname), seed=seed,
                 uv=inp['uv'], pose=inp['pose'], intrinsics=inp['intrinsics'], object_mask=om,
                 ray_dirs=dirs.numpy(), cam_loc=cam_loc.numpy(),
                 points=pts, mask=mask, dists=dists, minsdf_steps=steps, rows=rows, **extra)


@contextlib.contextmanager
def env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    os.environ.update({k: str(v) for k, v in kw.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def g_trace_mlp(W, B, P, seed, render=0):
    """render=1: the reference's rendering variant of the tracer (IDR_USE_ENV=1 IDR_RENDER=1 -> dist_clip 0.05, 40 sphere-tracing iterations,
    ray_tracing.py:127-131; eval.py / IDR_RENDER runs) -> trace_mlp_w<W>_<mode>_render."""
    m, sd = build_model(W, seed)
    tr = synth.model_conf(W)['ray_tracer']
    inp, _ = synth.make_batch(B, P, 0, seed=seed, focal_scale=1.4, with_features=False)
    net = m.implicit_network
    net.eval()
    with torch.no_grad():
        dirs, cam_loc = rend_util.get_camera_params(T(inp['uv']), T(inp['pose']), T(inp['intrinsics']))
    om = np.ones((B * P,), dtype=bool)
    for training in (False, True):
        with (env(IDR_USE_ENV=1, IDR_RENDER=1) if render else contextlib.nullcontext()):
            pts, mask, dists, steps, rows, extra = run_tracer(lambda x: net(x)[:, 0], cam_loc, T(om), dirs, training,
                                                       seed + 5, margins=True, **tr)
        with torch.no_grad():
            sdf_at = net(T(pts))[:, 0].numpy()
        if render:
            extra['dist_clip'] = np.float32(0.05)
            extra['sphere_tracing_iters'] = np.int32(40)
        save('trace_mlp_w%d_%s%s' % (W, 'train' if training else 'eval', '_render' if render else ''), W=W, seed=seed, B=B, P=P,
             focal_scale=1.4, ray_dirs=dirs.numpy(), cam_loc=cam_loc.numpy(), points=pts, mask=mask,
             dists=dists, sdf_at_points=sdf_at, minsdf_steps=steps, rows=rows, **extra,
             checksum=synth.state_checksum(sd))


def g_sample_network(seed):
    rs = np.random.RandomState(seed)
    n = 200
    a = {k: rs.normal(size=s).astype(np.float32) for k, s in dict(
        surface_output=(n, 1), surface_sdf_values=(n, 1), surface_points_grad=(n, 3), surface_dists=(n, 1),
        surface_cam_loc=(n, 3), surface_ray_dirs=(n, 3)).items()}
    out = SampleNetwork()(*[T(a[k]) for k in ('surface_output', 'surface_sdf_values', 'surface_points_grad',
                                               'surface_dists', 'surface_cam_loc', 'surface_ray_dirs')]).numpy()
    save('sample_network', out=out, **a)


SCENE = dict(size=2.6, center=(0.1, -0.2, 0.3), feat_hw=(60, 80), focal_scale=1.4)


def g_idr(W, B, P, V, seed, tp, name=None, skip_in=(4,), smooth=None, use_invalid=False, use_mask=False, enable_rgb=True):
    """smooth: conf.smooth(tp) of the depth term (loss.py:57-58: SmoothL1 instead of L1; None in the shipped conf) -- set on the reference's conf module for this fixture.
    use_mask: conf.use_mask = True (idr.py:186: the tracer and the rgb term see input['object_mask']) with a random 70 % object mask (stored in the fixture).
    enable_rgb=False: conf.enable_rgb off (loss.py:184-187: rgb_loss = zeros(1))."""
    import model.loss as ref_loss
    old_smooth, old_ui = ref_loss.conf.smooth, ref_loss.conf.use_invalid
    old_um, old_er = ref_loss.conf.use_mask, ref_loss.conf.enable_rgb
    ref_loss.conf.use_mask, ref_loss.conf.enable_rgb = bool(use_mask), bool(enable_rgb)   # (model.conf: the module idr.py and loss.py share)
    ref_loss.conf.use_invalid = bool(use_invalid)
    if smooth is not None:
        ref_loss.conf.smooth = lambda tp_: smooth
    m, sd = build_model(W, seed, skip_in=skip_in)
    inp, gt = synth.make_batch(B, P, V, seed=seed, **SCENE)
    if use_invalid:                                                            # depth maps with holes: that is where carving_t and carving_t2 differ
        inp['depths'] = gt['depths'] = synth.make_depth_maps(inp['depth_cams'], SCENE['size'], SCENE['center'], seed=seed, hole_frac=0.3)
    if use_mask:
        inp['object_mask'] = np.random.RandomState(seed + 17).uniform(size=inp['object_mask'].shape) < 0.7
    m.train()
    torch.manual_seed(seed + 5)
    mi = {k: T(v) for k, v in inp.items()}
    with torch.no_grad():
        dirs, cam_loc = rend_util.get_camera_params(mi['uv'], mi['pose'], mi['intrinsics'])
    rec = MarginRecorder(m.ray_tracer, cam_loc, dirs)             # per-ray decision margins of the tracer call inside forward
    rt_forward = m.ray_tracer.forward
    m.ray_tracer.forward = lambda sdf, **k: rt_forward(sdf=rec.wrap(sdf), **k)
    with quiet():
        out = m(mi, tp)
    res = dict(rec.arrays())
    for k, v in out.items():
        res['out_' + k] = v.detach().numpy()
    # loss + gradients
    loss_fn = IDRLoss()
    gtt = {k: T(v) for k, v in gt.items()}
    with quiet():
        lo = loss_fn(out, gtt, tp, B)
        feat_alone = loss_fn.get_feat_loss_corr(out['diff_surf_pts'], None, gtt['feat'], gtt['cam'], gtt['feat_src'],
                                                gtt['src_cams'], gtt['size'], gtt['center'],
                                                out['network_object_mask'], out['object_mask'])
    for k, v in lo.items():
        res['loss_' + k] = v.detach().numpy().reshape(-1)[0]
    res['feat_loss_alone'] = feat_alone.detach().numpy().reshape(-1)[0]
    m.zero_grad()
    lo['loss'].backward()
    rs = np.random.RandomState(1)
    for k, p in m.named_parameters():
        g = p.grad.detach().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        res['gnorm_' + k] = np.linalg.norm(g.astype(np.float64))
        idx = rs.randint(0, g.size, size=8)
        res['gidx_' + k] = idx
        res['gval_' + k] = g.reshape(-1)[idx]
    ref_loss.conf.smooth, ref_loss.conf.use_invalid = old_smooth, old_ui
    ref_loss.conf.use_mask, ref_loss.conf.enable_rgb = old_um, old_er
    if use_mask:
        res['use_mask'] = np.int32(1)
        res['in_object_mask'] = inp['object_mask']
    if not enable_rgb:
        res['enable_rgb'] = np.int32(0)
    if use_invalid:
        res['use_invalid'] = np.int32(1)
        res['depth_hole_frac'] = np.float32(0.3)
    if smooth is not None:
        res['smooth'] = np.float32(smooth)
    # the eikonal points drawn inside forward (torch CPU generator), for implementations that take them as input
    save(name or 'idr_w%d_tp%s' % (W, str(tp).replace('.', '')), W=W, B=B, P=P, V=V, seed=seed, tp=tp,
         scene_size=SCENE['size'], scene_center=np.array(SCENE['center']), feat_hw=np.array(SCENE['feat_hw']),
         focal_scale=SCENE['focal_scale'], checksum=synth.state_checksum(sd), **res)


IDR_FIXTURES = {   # name: (W, B, P, V, seed, tp, skip_in) of the g_idr fixtures (the rendering network's forward does not depend on their other options)
    'idr_w64_tp03': (64, 2, 256, 3, 0, 0.3, (4,)), 'idr_w64_tp06': (64, 2, 256, 3, 0, 0.6, (4,)), 'idr_w256_tp03': (256, 2, 128, 2, 0, 0.3, (4,)),
    'idr_c1': (256, 1, 512, 4, 0, 0.3, (4,)), 'idr_c2': (256, 8, 256, 4, 0, 0.3, (4,)), 'idr_c3': (256, 8, 1024, 8, 0, 0.3, (4,)),
    'idr_c5share': (256, 8, 512, 8, 0, 0.3, (4,)), 'idr_w512': (512, 8, 128, 2, 0, 0.3, (4,)), 'idr_w64_skips36': (64, 2, 256, 3, 0, 0.3, (3, 6)),
    'idr_w64_skip8': (64, 2, 256, 3, 0, 0.3, (8,)), 'idr_w64_smooth': (64, 2, 256, 3, 0, 0.3, (4,)), 'idr_w64_invalid': (64, 2, 256, 3, 0, 0.3, (4,)),
    'idr_w64_usemask': (64, 2, 256, 3, 0, 0.3, (4,)), 'idr_w64_norgb': (64, 2, 256, 3, 0, 0.3, (4,)),
}


def g_idr_eval(W, B, P, seed, render=0, name=None):
    """IDRNetwork.eval() forward of the reference (idr.py:179-322 with self.training False: tracer in eval mode, rgb_values = ones outside the surface mask, no
    eikonal keys, grad_theta None) as evaluation/eval.py:145-151 calls it (`model(s)`: train_progress None); render=1: under IDR_USE_ENV=1 IDR_RENDER=1 (40 iterations,
    dist_clip 0.05)."""
    m, sd = build_model(W, seed)
    inp, _ = synth.make_batch(B, P, 0, seed=seed, focal_scale=1.4, with_features=False)
    m.eval()
    mi = {k: T(v) for k, v in inp.items()}
    with (env(IDR_USE_ENV=1, IDR_RENDER=1) if render else contextlib.nullcontext()), quiet():
        out = m(mi)                                              # (not under no_grad: the reference's normals come from autograd.grad, idr.py:96-107)
    assert out['grad_theta'] is None and 'eikonal_output' not in out
    res = {'out_' + k: v.detach().numpy() for k, v in out.items() if v is not None}
    save(name or 'idr_eval_w%d%s' % (W, '_render' if render else ''), W=W, B=B, P=P, seed=seed, focal_scale=1.4, render=np.int32(render),
         checksum=synth.state_checksum(sd), out_keys=np.array(sorted(out.keys())), **res)


def g_idr_relu_margins():
    """Decision margins of the RENDERING network's ReLUs in the reference forward of every g_idr fixture: min |pre-activation| per hidden layer over the
    hit rows (idr.py:160-165).  A pre-activation within the forward noise of zero (1e-6 .. 1e-5: the features and normals it is computed from agree with the
    reference to ~2e-6) may take the other branch in an implementation with another fp32 summation order: one row's contribution to that unit's gradient
    flips, which moves the entries of the (small) rendering-network gradients of that layer and the layers below it by up to ~1e-2 of their scale -- a tie, like
    the tracer's recorded margins.  tests/test_gpu_idr.py widens the sampled-entry tolerance of exactly those layers when a margin is below 1e-5."""
    res = {}
    for name, (W, B, P, V, seed, tp, skip_in) in IDR_FIXTURES.items():
        m, sd = build_model(W, seed, skip_in=skip_in)
        inp, gt = synth.make_batch(B, P, V, seed=seed, **SCENE)
        m.train()
        torch.manual_seed(seed + 5)
        rn = m.rendering_network
        mins = {}
        hooks = []
        for l in range(rn.num_layers - 2):                       # the Linears followed by a ReLU
            hooks.append(getattr(rn, 'lin%d' % l).register_forward_hook(
                lambda mod, i, o, l=l: mins.__setitem__(l, min(mins.get(l, np.inf), float(o.detach().abs().min())))))
        with quiet():
            m({k: T(v) for k, v in inp.items()}, tp)
        for h in hooks:
            h.remove()
        res[name] = np.array([mins[l] for l in range(rn.num_layers - 2)], np.float64)
        print('%-18s min |pre-activation| per rendering layer: %s' % (name, ' '.join('%.2e' % v for v in res[name])))
    save('idr_relu_margins', **res)


def g_feat(seed, B=2, P=300, V=3, name='feat_corr'):
    """get_feat_loss_corr alone on fixed points + d loss / d points (V = 3 / 4 / 8 source views)."""
    inp, gt = synth.make_batch(B, P, V, seed=seed, **SCENE)
    rs = np.random.RandomState(seed + 9)
    hits = rs.uniform(size=(B * P,)) < 0.7
    if B > 2:
        hits[P:2 * P] = False                                   # one view without any hit (loss.py:117-118 path)
    n = int(hits.sum())
    d = rs.normal(size=(n, 3))
    pts = (0.6 * d / np.linalg.norm(d, axis=1, keepdims=True) + 0.03 * rs.normal(size=(n, 3))).astype(np.float32)
    p = T(pts).requires_grad_(True)
    gtt = {k: T(v) for k, v in gt.items()}
    with quiet():
        loss = IDRLoss().get_feat_loss_corr(p, None, gtt['feat'], gtt['cam'], gtt['feat_src'], gtt['src_cams'],
                                            gtt['size'][:1], gtt['center'][:1], T(hits), T(np.ones_like(hits)))
    loss.backward()
    save(name, seed=seed, B=B, P=P, V=V, hits=hits, points=pts, loss=loss.item(), dpoints=p.grad.numpy(),
         scene_size=SCENE['size'], scene_center=np.array(SCENE['center']), feat_hw=np.array(SCENE['feat_hw']),
         focal_scale=SCENE['focal_scale'])


def g_carve(seed, use_invalid=False, name='carve'):
    """carving_t2 (my_utils.py:269-331) + get_depth_loss (loss.py:37-63) on bumpy depth maps with holes, a depth step and per-view scale
    errors: points inside / outside the surface, near it (views disagree: out_thresh_perc voting) and outside every frustum."""
    from utils.my_utils import carving_t, carving_t2
    import model.loss as ref_loss
    old_ui = ref_loss.conf.use_invalid
    ref_loss.conf.use_invalid = bool(use_invalid)                                  # conf.use_invalid: carving_t instead of carving_t2 (loss.py:43-46)
    B, hw, M = 5, (48, 64), 4000
    inp, _ = synth.make_batch(B, 8, 1, seed=seed, size=SCENE['size'], center=SCENE['center'], feat_hw=hw, focal_scale=1.4, with_features=False)
    depths = synth.make_depth_maps(inp['depth_cams'], SCENE['size'], SCENE['center'], seed=seed)
    rs = np.random.RandomState(seed + 41)
    pts = rs.uniform(-1.3, 1.3, size=(M, 3))
    d = rs.normal(size=(M // 2, 3))
    pts[: M // 2] = d / np.linalg.norm(d, axis=1, keepdims=True) * (0.6 + 0.05 * rs.normal(size=(M // 2, 1)))    # half of them around the surface
    pts[-200:] *= 4.0                                                                                             # far outside every frustum
    pts = pts.astype(np.float32)
    eik_out = (0.3 * rs.normal(size=(1, M))).astype(np.float32)
    size, center = T(inp['size'])[:1], T(inp['center'])[:1]
    hom = torch.cat([T(pts), torch.ones(M, 1)], -1).view(1, M, 4, 1)
    world = hom.clone()
    world[:, :, :3, 0] = world[:, :, :3, 0] / 2 * size.view(1, 1, 1) + center.view(1, 1, 3)
    dp, cp = T(depths).permute(1, 0, 2, 3, 4), T(inp['depth_cams']).permute(1, 0, 2, 3, 4)
    dist, occ, in_range = (carving_t if use_invalid else carving_t2)(world, dp, cp, out_thresh_perc=1 / 8)
    res = dict(seed=seed, depths=depths, depth_cams=inp['depth_cams'], size=inp['size'], center=inp['center'], points=pts, eik_out=eik_out,
               dist=dist.numpy()[0], occ=occ.numpy()[0], in_range=in_range.numpy()[0])
    for tag, (fa, na) in dict(a=(1.0, 1.0), b=(1.0, 0.1), c=(0.5, 0.01)).items():
        loss = IDRLoss().get_depth_loss(hom.clone(), T(eik_out), T(depths), T(inp['depth_cams']), size, center, 0.25, fa, 0.1, na, None)
        res['loss_' + tag] = loss.item()
        res['att_' + tag] = np.array([fa, na])
    ref_loss.conf.use_invalid = old_ui
    res['use_invalid'] = np.int32(1 if use_invalid else 0)
    save(name, **res)


def g_idr_phase0(W, B, P, V, seed, tp):
    """Phase 0 (tp < 1/6): depth-surface sampling on (idr.py:226-247), every point group in the depth and eikonal terms, rgb gradient
    through the features only (idr.py:331-334).  The dsurf samples the reference drew (np.random.choice / rand_like) are stored: the
    build's own sampler uses another RNG, so its test injects these points."""
    m, sd = build_model(W, seed)
    inp, gt = synth.make_batch(B, P, V, seed=seed, **SCENE)
    inp['depths'] = gt['depths'] = synth.make_depth_maps(inp['depth_cams'], SCENE['size'], SCENE['center'], seed=seed, hole_frac=0.05)
    m.train()
    torch.manual_seed(seed + 5)
    np.random.seed(seed + 6)
    mi = {k: T(v) for k, v in inp.items()}
    with quiet():
        out = m(mi, tp)
    res = {('out_' + k): v.detach().numpy().copy() for k, v in out.items()}     # copies: the loss rescales eikonal_points_hom in place
    N = int((out['network_object_mask'] & out['object_mask']).sum())
    R = B * P
    hom = res['out_eikonal_points_hom'][0, :, :3, 0]
    res['dsurf_on'] = hom[N + R // 2: N + R].copy()
    res['dsurf_jitter'] = hom[N + R:].copy()
    assert res['dsurf_jitter'].shape[0] == R // 2
    loss_fn = IDRLoss()
    gtt = {k: T(v) for k, v in gt.items()}
    with quiet():
        lo = loss_fn(out, gtt, tp, B)
    for k, v in lo.items():
        res['loss_' + k] = (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v, np.float32)).reshape(-1)[0]
    m.zero_grad()
    lo['loss'].backward()
    rs = np.random.RandomState(1)
    for k, p in m.named_parameters():
        g = p.grad.detach().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        res['gnorm_' + k] = np.linalg.norm(g.astype(np.float64))
        idx = rs.randint(0, g.size, size=8)
        res['gidx_' + k] = idx
        res['gval_' + k] = g.reshape(-1)[idx]
    save('idr_w%d_phase0' % W, W=W, B=B, P=P, V=V, seed=seed, tp=tp, scene_size=SCENE['size'], scene_center=np.array(SCENE['center']),
         feat_hw=np.array(SCENE['feat_hw']), focal_scale=SCENE['focal_scale'], checksum=synth.state_checksum(sd), **res)


def g_sdf_bwd(W, n, seed, skip_in=(4,), name=None):
    """Pins the (double) backward: L = sum(out*dy) + sum(grad*dn) through ImplicitNetwork.forward + .gradient
    (idr.py:77-107) -> d/d{weight_g, weight_v, bias} of every layer and d/dx.  skip_in: several skip connections (idr.py:46,86)."""
    m, sd = build_model(W, seed, skip_in=skip_in)
    net = m.implicit_network
    net.train()
    rs = np.random.RandomState(seed + 21)
    x = rs.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    dy = (rs.normal(size=(n, 1 + 1 + synth.FEAT)) * 0.1).astype(np.float32)
    dn = rs.normal(size=(n, 3)).astype(np.float32)
    xt = T(x).clone().requires_grad_(True)
    out = net(xt)
    g = net.gradient(xt)[:, 0, :]
    L = (out * T(dy)).sum() + (g * T(dn)).sum()
    params = [p for _, p in net.named_parameters()]
    grads = torch.autograd.grad(L, [xt] + params)
    res = dict(W=W, seed=seed, x=x, dy=dy, dn=dn, dx=grads[0].numpy(), checksum=synth.state_checksum(sd), skip_in=np.array(skip_in),
               out=out.detach().numpy(), grad=g.detach().numpy())
    for (k, _), gr in zip(net.named_parameters(), grads[1:]):
        res['d_' + k] = gr.numpy()
    # first-order only variant (dn = 0), dx through the value chain alone
    xt2 = T(x).clone().requires_grad_(True)
    out2 = net(xt2)
    res['dx_value_only'] = torch.autograd.grad((out2 * T(dy)).sum(), [xt2])[0].numpy()
    save(name or 'sdf_bwd_w%d' % W, **res)


def g_render_bwd(W, n, seed):
    m, sd = build_model(W, seed)
    net = m.rendering_network
    rs = np.random.RandomState(seed + 31)
    pts = rs.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    nrm = rs.normal(size=(n, 3)).astype(np.float32)
    view = rs.normal(size=(n, 3)).astype(np.float32)
    view /= np.linalg.norm(view, axis=1, keepdims=True)
    feat = rs.normal(size=(n, synth.FEAT)).astype(np.float32)
    drgb = rs.normal(size=(n, 3)).astype(np.float32)
    ins = [T(a).clone().requires_grad_(True) for a in (pts, nrm, feat)]
    rgb = net(ins[0], ins[1], T(view), ins[2])
    params = [p for _, p in net.named_parameters()]
    grads = torch.autograd.grad((rgb * T(drgb)).sum(), ins + params)
    res = dict(W=W, seed=seed, points=pts, normals=nrm, view=view, feat=feat, drgb=drgb, rgb=rgb.detach().numpy(),
               dpoints=grads[0].numpy(), dnormals=grads[1].numpy(), dfeat=grads[2].numpy(), checksum=synth.state_checksum(sd))
    for (k, _), gr in zip(net.named_parameters(), grads[3:]):
        res['d_' + k] = gr.numpy()
    save('render_bwd_w%d' % W, **res)


def g_train(W, B, P, V, seed, tp, steps, lr):
    """A few real optimisation steps of the reference loop (idr_train.py:283-302): zero_grad, forward, loss, backward, all_norm,
    clip_grad_norm_(grad_cap), Adam.step -- per-step losses, gradient norm, hit count and the parameter norms at the end."""
    from model import conf as rconf
    m, sd = build_model(W, seed)
    inp, gt = synth.make_batch(B, P, V, seed=seed, **SCENE)
    m.train()
    mi, gtt = {k: T(v) for k, v in inp.items()}, {k: T(v) for k, v in gt.items()}
    loss_fn = IDRLoss()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    losses, gnorms, hits = [], [], []
    for it in range(steps):
        torch.manual_seed(seed + 100 + it)
        opt.zero_grad()
        with quiet():
            out = m(mi, tp)
            lo = loss_fn(out, dict(gtt), tp, B)
        lo['loss'].backward()
        all_norm = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).norm()
        if rconf.phase[0] <= tp and rconf.enable_grad_cap:
            torch.nn.utils.clip_grad_norm_(m.parameters(), rconf.grad_cap(tp))
        opt.step()
        losses.append([float(lo[k].reshape(-1)[0]) for k in ('loss', 'rgb_loss', 'eikonal_loss', 'depth_loss', 'feat_loss', 'surf_loss')])
        gnorms.append(float(all_norm)); hits.append(int((out['network_object_mask'] & out['object_mask']).sum()))
    pn = {('pnorm_' + k): float(p.detach().double().norm()) for k, p in m.named_parameters()}
    save('train%d_w%d' % (steps, W), W=W, B=B, P=P, V=V, seed=seed, tp=tp, steps=steps, lr=lr, losses=np.array(losses), gnorms=np.array(gnorms),
         hits=np.array(hits), scene_size=SCENE['size'], scene_center=np.array(SCENE['center']), feat_hw=np.array(SCENE['feat_hw']),
         focal_scale=SCENE['focal_scale'], grad_cap=float(rconf.grad_cap(tp)), checksum=synth.state_checksum(sd), **pn)


def g_dsurf(seed):
    """Phase-0 depth-surface points (idr.py:234-238): every depth pixel unprojected with the reference's my_utils helpers + the
    in-box tests of idr.py:242 (on-surface points; the jitter is RNG-driven and checked distributionally in the GPU test)."""
    from utils.my_utils import get_pixel_grids, idx_cam2world, idx_img2cam
    B, hw = 3, (24, 32)
    inp, _ = synth.make_batch(B, 8, 1, seed=seed, feat_hw=hw, with_features=False)
    rs = np.random.RandomState(seed + 77)
    depths = inp['depths'].copy()                                              # [B,1,1,h,w]
    depths *= rs.uniform(0.6, 1.4, size=depths.shape).astype(np.float32)       # a bumpy surface: some points leave the box
    depths[rs.rand(*depths.shape) < 0.3] = 0.0                                 # holes (depth <= 0 is invalid)
    d, c = T(depths), T(inp['depth_cams'])
    center, size = T(inp['center'])[:1], T(inp['size'])[:1]
    dp, cp = [a.view(-1, *a.size()[2:]) for a in (d, c)]
    hom = idx_cam2world(idx_img2cam(get_pixel_grids(*d.size()[-2:], cuda=False).unsqueeze(0), dp, cp), cp)   # N h w 4 1
    pts_all = (hom[..., :3, 0] - center) / size * 2                            # N h w 3 (normalised, every pixel)
    valid = dp[:, 0] > 0
    bb = 1.0
    inbound = ((pts_all.abs() < bb).float().sum(-1) > 2.9) & valid
    save('dsurf_unproject', seed=seed, depths=depths, depth_cams=inp['depth_cams'], size=inp['size'], center=inp['center'], bb=bb,
         pts_norm=pts_all.numpy(), valid=valid.numpy(), inbound=inbound.numpy())


# fixtures `--check NAME ...` can regenerate, and the call that writes each
RECIPES = {
    'idr_w64_tp03': lambda: g_idr(64, 2, 256, 3, 0, 0.3), 'idr_w64_tp06': lambda: g_idr(64, 2, 256, 3, 0, 0.6), 'idr_w256_tp03': lambda: g_idr(256, 2, 128, 2, 0, 0.3),
    'sdf_w64': lambda: g_sdf(64, 1000, 0), 'render_w64': lambda: g_render(64, 300, 0), 'rays': lambda: g_rays(0), 'sample_network': lambda: g_sample_network(0),
    'feat_corr': lambda: g_feat(0), 'carve': lambda: g_carve(0), 'sdf_bwd_w64': lambda: g_sdf_bwd(64, 150, 0), 'render_bwd_w64': lambda: g_render_bwd(64, 150, 0),
    'idr_w64_usemask': lambda: g_idr(64, 2, 256, 3, 0, 0.3, 'idr_w64_usemask', (4,), None, False, True),
    'idr_w64_norgb': lambda: g_idr(64, 2, 256, 3, 0, 0.3, 'idr_w64_norgb', (4,), None, False, False, False),
    'idr_eval_w64': lambda: g_idr_eval(64, 2, 300, 0), 'idr_eval_w64_render': lambda: g_idr_eval(64, 2, 300, 0, 1), 'idr_eval_w256': lambda: g_idr_eval(256, 2, 200, 0),
    'dsurf_unproject': lambda: g_dsurf(0),
}


def check(names):
    """`make_golden.py --check NAME ...`: regenerate the named fixtures into a scratch directory and compare them with the committed files array by array,
    bit for bit (a fixture that drifted from its generator -- arrays missing, values changed -- fails here).  -> number of differing fixtures."""
    import tempfile
    global OUT_DIR
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        OUT_DIR = tmp
        try:
            for n in names:
                with quiet():
                    RECIPES[n]()
                new, old = np.load(os.path.join(tmp, n + '.npz'), allow_pickle=False), np.load(os.path.join(HERE, n + '.npz'), allow_pickle=False)
                diff = sorted(set(new.files) ^ set(old.files)) + [k for k in new.files if k in old.files and not (
                    new[k].shape == old[k].shape and new[k].dtype == old[k].dtype and np.array_equal(new[k], old[k], equal_nan=new[k].dtype.kind == 'f'))]
                print('%-28s %s' % (n, 'identical (%d arrays)' % len(new.files) if not diff else 'DIFFERS: ' + ', '.join(diff)))
                bad += bool(diff)
        finally:
            OUT_DIR = HERE
    return bad


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--check':
        sys.exit(1 if check(sys.argv[2:]) else 0)
    if len(sys.argv) > 1:                                                       # python make_golden.py g_dsurf 0  (one fixture)
        def _arg(v):
            for conv in (int, float):
                try:
                    return conv(v)
                except ValueError:
                    pass
            return v
        globals()[sys.argv[1]](*[_arg(v) for v in sys.argv[2:]])
        sys.exit(0)
    g_dsurf(0)
    g_train_default = lambda: g_train(64, 2, 256, 3, 0, 0.3, 4, 1e-3)
    g_train_default()
    g_sdf(64, 1000, 0)
    g_sdf(256, 256, 0)
    g_render(64, 300, 0)
    g_rays(0)
    g_trace_analytic(0)
    g_trace_mlp(64, 4, 1024, 0)
    g_trace_mlp(256, 2, 512, 0)
    g_sample_network(0)
    g_feat(0)
    g_sdf_bwd(64, 150, 0)
    g_render_bwd(64, 150, 0)
    g_idr(64, 2, 256, 3, 0, 0.3)
    g_idr(64, 2, 256, 3, 0, 0.6)
    g_idr(256, 2, 128, 2, 0, 0.3)
    g_feat(0, 8, 200, 4, 'feat_corr_v4')
    g_feat(0, 8, 160, 8, 'feat_corr_v8')
    g_carve(0)
    g_idr_phase0(64, 3, 128, 2, 0, 0.1)
    g_idr(256, 8, 256, 4, 0, 0.3, 'idr_c2')
    g_idr(256, 8, 1024, 8, 0, 0.3, 'idr_c3')
    g_sdf_bwd(64, 150, 0, (3, 6), 'sdf_bwd_w64_skips36')                        # several skip connections (idr.py:46,86)
    g_idr(64, 2, 256, 3, 0, 0.3, 'idr_w64_skips36', (3, 6))
    g_idr(256, 8, 512, 8, 0, 0.3, 'idr_c5share')                                # one GPU's share of BASELINE configs[4] (4096 rays, V = 8)
    g_idr(256, 1, 512, 4, 0, 0.3, 'idr_c1')                                     # BASELINE configs[0] at its own shape: B = 1, 512 rays, V = 4
    # the reference's SHIPPED configuration: 8x512 SDF net, 4x512 rendering net (confs/mvsdf_dtu.conf:24,35), num_src = 2 (scene_dataset.py:104)
    g_sdf(512, 256, 0)
    g_trace_mlp(512, 2, 512, 0)
    g_idr(512, 8, 128, 2, 0, 0.3, 'idr_w512')
    g_idr(64, 2, 256, 3, 0, 0.3, 'idr_w64_smooth', (4,), 0.05)
    g_carve(0, True, 'carve_invalid')                                           # conf.use_invalid: carving_t (loss.py:43-44)
    g_idr(64, 2, 256, 3, 0, 0.3, 'idr_w64_invalid', (4,), None, True)                  # conf.smooth = 0.05: the SmoothL1 depth term (loss.py:57-58), reachable through IDR_CONF
    g_sdf_bwd(64, 150, 0, (8,), 'sdf_bwd_w64_skip8')                            # a skip connection into the LAST Linear (idr.py:46-49,86)
    g_sdf_bwd(64, 150, 0, (4, 8), 'sdf_bwd_w64_skips48')
    g_idr(64, 2, 256, 3, 0, 0.3, 'idr_w64_skip8', (8,))
    g_idr(64, 2, 256, 3, 0, 0.3, 'idr_w64_usemask', (4,), None, False, True)          # conf.use_mask = True (idr.py:186) with a random object mask
    g_idr(64, 2, 256, 3, 0, 0.3, 'idr_w64_norgb', (4,), None, False, False, False)     # conf.enable_rgb = False (loss.py:184-187)
    # the rendering variant of the tracer (IDR_RENDER: 40 iterations, dist_clip 0.05, ray_tracing.py:127-131) and IDRNetwork in eval mode (eval.py:145-151)
    g_trace_mlp(64, 4, 1024, 0, 1)
    g_trace_mlp(256, 2, 512, 0, 1)
    g_idr_eval(64, 2, 300, 0)
    g_idr_eval(64, 2, 300, 0, 1)
    g_idr_eval(256, 2, 200, 0)
    g_idr_relu_margins()
