"""Deterministic synthetic scenes (weights, cameras, features, batches).

Shared by the golden-vector generator (tests/golden/make_golden.py, which feeds
the SAME arrays to the PyTorch reference), the parity tests and bench.py, so
that "identical ray batches" is true by construction.  Everything is drawn from
``numpy.random.RandomState`` (a frozen, version-stable stream) in float64 and
cast to float32 -- never from torch RNGs -- so the GPU box regenerates the
exact weights/batches the fixtures were made with (weights are not committed,
only their checksums).

Recipe = SURVEY.md section 8(d):
  * SDF net: geometric init (a radius-`bias` sphere; reference idr.py:53-68)
    + 0.02*mean|v|*N(0,1) on every weight_v  -> bumpy closed surface.
  * rendering net: nn.Linear default init (U(-1/sqrt(in), 1/sqrt(in))).
  * cameras on a circle looking at the origin; MVS-world = x/2*size + center.
  * features: shared base + smoothed noise (i.i.d. noise zeroes the loss mask).
"""
from collections import OrderedDict

import numpy as np

PE_SDF = 6      # multires (mvsdf_dtu.conf:29)
PE_VIEW = 4     # multires_view (mvsdf_dtu.conf:38)
FEAT = 256      # feature_vector_size (mvsdf_dtu.conf:19)


def sdf_layer_dims(W, n_hidden=8, multires=PE_SDF, feat=FEAT, skip_in=(4,)):
    """[(in, out)] per Linear of the SDF net (reference idr.py:33-51)."""
    d0 = 3 + 6 * multires if multires > 0 else 3
    dims = [d0] + [W] * n_hidden + [1 + 1 + feat]
    out = []
    for l in range(len(dims) - 1):
        o = dims[l + 1] - d0 if (l + 1) in skip_in else dims[l + 1]
        out.append((dims[l], o))
    return out


def render_layer_dims(W, n_hidden=4, multires_view=PE_VIEW, feat=FEAT):
    """[(in, out)] per Linear of the rendering net, mode 'idr' (idr.py:121-131)."""
    d0 = 9 + feat + (6 * multires_view if multires_view > 0 else 0)
    dims = [d0] + [W] * n_hidden + [3]
    return [(dims[l], dims[l + 1]) for l in range(len(dims) - 1)]


def make_state_dict(W, seed=0, noise=0.02, bias=0.6, n_hidden=8, n_hidden_r=4, skip_in=(4,)):
    """state_dict (numpy float32) in the reference key layout
    implicit_network.lin{l}.{bias,weight_g,weight_v}, rendering_network.lin{l}.*"""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    dims = sdf_layer_dims(W, n_hidden, skip_in=tuple(skip_in))
    d0 = dims[0][0]
    L = len(dims)
    for l, (i, o) in enumerate(dims):
        if l == L - 1:
            w = rs.normal(np.sqrt(np.pi) / np.sqrt(i), 1e-4, size=(o, i))
            b = np.full((o,), -bias)
            if l in skip_in:                                     # a skip into the last Linear: small PE columns keep the synthetic surface near the sphere
                w[:, -d0:] = rs.normal(0.0, 0.01, size=(o, d0))
        elif l == 0:
            w = np.zeros((o, i))
            w[:, :3] = rs.normal(0.0, np.sqrt(2) / np.sqrt(o), size=(o, 3))
            b = np.zeros((o,))
        elif l in skip_in:
            w = rs.normal(0.0, np.sqrt(2) / np.sqrt(o), size=(o, i))
            w[:, -(d0 - 3):] = 0.0
            b = np.zeros((o,))
        else:
            w = rs.normal(0.0, np.sqrt(2) / np.sqrt(o), size=(o, i))
            b = np.zeros((o,))
        g = np.sqrt((w * w).sum(1, keepdims=True))
        v = w + noise * np.abs(w).mean() * rs.normal(size=w.shape)
        sd['implicit_network.lin%d.bias' % l] = b.astype(np.float32)
        sd['implicit_network.lin%d.weight_g' % l] = g.astype(np.float32)
        sd['implicit_network.lin%d.weight_v' % l] = v.astype(np.float32)
    for l, (i, o) in enumerate(render_layer_dims(W, n_hidden_r)):
        k = 1.0 / np.sqrt(i)
        w = rs.uniform(-k, k, size=(o, i))
        b = rs.uniform(-k, k, size=(o,))
        sd['rendering_network.lin%d.bias' % l] = b.astype(np.float32)
        sd['rendering_network.lin%d.weight_g' % l] = np.sqrt((w * w).sum(1, keepdims=True)).astype(np.float32)
        sd['rendering_network.lin%d.weight_v' % l] = w.astype(np.float32)
    return sd


def state_checksum(sd):
    """Cheap fingerprint stored in fixtures to prove weights were regenerated identically."""
    return np.array([float(np.float64(v).sum()) for v in sd.values()]
                    + [float(np.abs(np.float64(v)).sum()) for v in sd.values()])


def model_conf(W, line_step_iters=3, n_hidden=8, n_hidden_r=4, skip_in=(4,)):
    """Plain-dict model config equal to the 'model' block of mvsdf_dtu.conf:18-58 at width W."""
    return dict(
        feature_vector_size=FEAT,
        implicit_network=dict(d_in=3, d_out=1, dims=[W] * n_hidden, geometric_init=True, bias=0.6,
                              skip_in=list(skip_in), weight_norm=True, multires=PE_SDF),
        rendering_network=dict(mode='idr', d_in=9, d_out=3, dims=[W] * n_hidden_r, weight_norm=True,
                               multires_view=PE_VIEW),
        ray_tracer=dict(object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5,
                        line_step_iters=line_step_iters, sphere_tracing_iters=10, n_steps=100,
                        n_secant_steps=8))


def _look_at(c):
    """camera-to-world rotation (columns = camera x,y,z axes in world), z looks at the origin."""
    z = -c / np.linalg.norm(c)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=1)


def _camera(angle, radius, height, size, center, img_wh, focal, feat_hw):
    """-> pose[4,4] (normalised c2w), K_img[4,4], mvs cam[2,4,4] at feature/depth resolution."""
    c = np.array([radius * np.cos(angle), radius * np.sin(angle), height])
    R = _look_at(c)
    pose = np.eye(4)
    pose[:3, :3] = R
    pose[:3, 3] = c
    Wi, Hi = img_wh
    K = np.eye(4)
    K[0, 0] = K[1, 1] = focal
    K[0, 2], K[1, 2] = Wi / 2.0, Hi / 2.0
    Hf, Wf = feat_hw
    s = Wf / float(Wi)
    cw = c / 2.0 * size + np.asarray(center, dtype=np.float64)      # camera centre in MVS world
    E = np.eye(4)
    E[:3, :3] = R.T
    E[:3, 3] = -R.T @ cw
    cam = np.zeros((2, 4, 4))
    cam[0] = E
    cam[1, 0, 0] = cam[1, 1, 1] = focal * s
    cam[1, 0, 2], cam[1, 1, 2] = Wf / 2.0, Hf / 2.0
    cam[1, 2, 2] = 1.0
    cam[1, 3, 3] = 1.0
    return pose, K, cam


def scale_cam(cam, s):
    """K -> s*K (focal + principal point), like reference my_utils.scale_camera (my_utils.py:31-61)."""
    out = np.array(cam, copy=True)
    out[..., 1, 0, 0] *= s
    out[..., 1, 1, 1] *= s
    out[..., 1, 0, 2] *= s
    out[..., 1, 1, 2] *= s
    return out


def make_features(n, C, H, W, seed):
    """feat = base[C] + 2.4*avgpool5x5(N(0,1)), one base shared by all n maps -> [n,C,H,W] float32."""
    rs = np.random.RandomState(seed)
    base = rs.normal(size=(1, C, 1, 1))
    z = rs.normal(size=(n, C, H + 4, W + 4)).astype(np.float32)
    acc = np.zeros((n, C, H, W), dtype=np.float32)
    for dy in range(5):
        for dx in range(5):
            acc += z[:, :, dy:dy + H, dx:dx + W]
    return (base + 2.4 * acc / 25.0).astype(np.float32)


def make_batch(B, P, V, seed=0, img_wh=(800, 600), focal_scale=2.2, radius=2.5, height=0.8,
               size=2.0, center=(0.0, 0.0, 0.0), feat_hw=(150, 200), C=32, depth_value=1.9,
               with_features=True):
    """Model input + ground truth for one step (numpy float32 / bool), shapes as the reference's
    collate_fn produces them (scene_dataset.py:189-242)."""
    rs = np.random.RandomState(seed + 1000)
    focal = focal_scale * img_wh[0]
    poses, Ks, cams, src = [], [], [], []
    for b in range(B):
        a = 2 * np.pi * b / max(B, 1) + 0.1
        p, K, c = _camera(a, radius, height, size, center, img_wh, focal, feat_hw)
        poses.append(p), Ks.append(K), cams.append(c)
        sv = []
        for j in range(V):
            off = (j // 2 + 1) * 0.12 * (1 if j % 2 == 0 else -1)
            sv.append(_camera(a + off, radius, height + 0.05 * (j - V / 2.0), size, center, img_wh, focal, feat_hw)[2])
        src.append(np.stack(sv) if V > 0 else np.zeros((0, 2, 4, 4)))
    uv = np.stack([rs.randint(0, img_wh[0], size=(B, P)), rs.randint(0, img_wh[1], size=(B, P))], -1)
    cams = np.stack(cams)
    inp = dict(
        intrinsics=np.stack(Ks).astype(np.float32),
        uv=uv.astype(np.float32),
        pose=np.stack(poses).astype(np.float32),
        object_mask=np.ones((B, P), dtype=bool),
        depths=np.full((B, 1, 1) + tuple(feat_hw), depth_value * size / 2.0, dtype=np.float32),
        depth_cams=cams[:, None].astype(np.float32),
        size=np.full((B,), size, dtype=np.float32),
        center=np.tile(np.asarray(center, dtype=np.float32)[None], (B, 1)),
    )
    gt = dict(
        rgb=rs.uniform(-1, 1, size=(B, P, 3)).astype(np.float32),
        depths=inp['depths'], depth_cams=inp['depth_cams'], size=inp['size'], center=inp['center'],
        cam=scale_cam(cams, 2).astype(np.float32),
        src_cams=scale_cam(np.stack(src), 2).astype(np.float32),
    )
    if with_features:
        f = make_features(B * (1 + V), C, feat_hw[0], feat_hw[1], seed + 2000)
        f = f.reshape(B, 1 + V, C, feat_hw[0], feat_hw[1])
        gt['feat'] = np.ascontiguousarray(f[:, 0])
        gt['feat_src'] = np.ascontiguousarray(f[:, 1:])
    return inp, gt


def make_depth_maps(depth_cams, size, center, seed=0, radius=0.6, hole_frac=0.15, bump=0.04, view_bias=0.03):
    """Non-trivial MVS depth maps [B,1,1,h,w] for the depth term / phase-0 sampling fixtures: camera-z depth of a bumpy sphere of
    normalised radius `radius` seen by each depth camera (`depth_cams` [B,1,2,4,4], intrinsics at depth-map resolution, pixel centres
    at +0.5 like my_utils.get_pixel_grids), 0 where the ray misses it, plus random holes (0), one depth step per view and a per-view
    scale error so that the views disagree about inside / outside near the surface (exercises carving_t2's voting)."""
    rs = np.random.RandomState(seed + 4000)
    cams = np.asarray(depth_cams, np.float64)[:, 0]
    B = cams.shape[0]
    h, w = int(round(2 * cams[0, 1, 1, 2])), int(round(2 * cams[0, 1, 0, 2]))
    xs, ys = np.meshgrid(np.arange(w) + 0.5, np.arange(h) + 0.5)
    out = np.zeros((B, 1, 1, h, w), np.float32)
    rw = radius * size / 2.0
    for b in range(B):
        E, K = cams[b, 0], cams[b, 1, :3, :3]
        Rm, t = E[:3, :3], E[:3, 3]
        o = -Rm.T @ t - np.asarray(center, np.float64)                      # camera centre relative to the sphere centre (world)
        d = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], -1) @ Rm   # world dirs with camera z = 1
        a = (d * d).sum(-1)
        bq = 2 * (d @ o)
        c = (o * o).sum() - rw * rw
        disc = bq * bq - 4 * a * c
        hit = disc > 0
        tz = np.where(hit, (-bq - np.sqrt(np.where(hit, disc, 0.0))) / (2 * a), 0.0)
        tz = tz * (1.0 + bump * np.sin(xs * 0.7 + b) * np.cos(ys * 0.5 - b)) * (1.0 + view_bias * (rs.uniform() - 0.5) * 2)
        y0, x0 = rs.randint(0, h // 2), rs.randint(0, w // 2)
        tz[y0:y0 + h // 4, x0:x0 + w // 4] += 0.12 * size / 2.0              # a depth step
        tz[rs.uniform(size=tz.shape) < hole_frac] = 0.0                       # holes
        out[b, 0, 0] = np.where(hit, tz, 0.0)
    return out
