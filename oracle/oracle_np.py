"""numpy restatement of the differentiable half of the MVSDF hot path (float64 by default).

TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Parity status: PINNED against golden vectors captured from the PyTorch reference (tests/golden/make_golden.py),
see tests/test_oracle_np_golden.py.  Citations are reference file:line (relative to /root/reference/code).

Covers: weight-norm fold (+backward), SDF value + normal forward and its first/second-order backward
(model/implicit_differentiable_renderer.py:77-107 + autograd; closed form of SURVEY.md App. E), rendering net
(idr.py:145-167), SampleNetwork (model/sample_network.py:10-20), feature-consistency loss with analytic d/dpoints
(model/loss.py:115-165, utils/my_utils.py:98-110,152-165), depth carving (utils/my_utils.py:269-331, model/loss.py:37-63)
and the remaining IDRLoss terms (model/loss.py:21-35,167-219).
"""
import numpy as np

SQRT2 = float(np.float32(np.sqrt(2.0)))     # torch divides a float32 tensor by fl32(np.sqrt(2)) (idr.py:87)


# ------------------------------------------------------------------------------------------------ weight norm
def fold(v, g):
    v = np.asarray(v, np.float64)
    return v * (np.asarray(g, np.float64).reshape(-1, 1) / np.linalg.norm(v, axis=1, keepdims=True))


def fold_backward(v, g, dW):
    """dW -> (dv, dg)   """
    v, dW = np.asarray(v, np.float64), np.asarray(dW, np.float64)
    g = np.asarray(g, np.float64).reshape(-1, 1)
    nrm = np.linalg.norm(v, axis=1, keepdims=True)
    vh = v / nrm
    dg = (dW * vh).sum(1, keepdims=True)
    dv = (g / nrm) * (dW - dg * vh)
    return dv, dg


class Net:
    def __init__(self, state, prefix, skip_layer=-1, multires=0):
        self.v, self.g, self.b, self.W = [], [], [], []
        l = 0
        while '%s.lin%d.weight_v' % (prefix, l) in state:
            self.v.append(np.asarray(state['%s.lin%d.weight_v' % (prefix, l)], np.float64))
            self.g.append(np.asarray(state['%s.lin%d.weight_g' % (prefix, l)], np.float64))
            self.b.append(np.asarray(state['%s.lin%d.bias' % (prefix, l)], np.float64))
            self.W.append(fold(self.v[-1], self.g[-1]))
            l += 1
        self.n_layers, self.multires = l, multires
        # skip_layer: one layer index (-1: none) or a sequence (skip_in, idr.py:46,86)
        self.skip_layers = tuple(skip_layer) if isinstance(skip_layer, (tuple, list)) else ((skip_layer,) if skip_layer >= 0 else ())
        self.skip_layer = self.skip_layers[0] if self.skip_layers else -1


def sdf_net(state, skip_in=(4,), multires=6):
    return Net(state, 'implicit_network', tuple(skip_in), multires)


def render_net(state):
    return Net(state, 'rendering_network')


# ------------------------------------------------------------------------------------------------ PE + activations
def pe(x, multires):
    """[x, sin(2^m x), cos(2^m x)]_m   (model/embedder.py:10-36)"""
    x = np.asarray(x, np.float64)
    out = [x]
    for m in range(multires):
        out += [np.sin(x * 2.0 ** m), np.cos(x * 2.0 ** m)]
    return np.concatenate(out, 1)


def softplus100(z):
    y = 100.0 * z
    return np.where(y > 20.0, z, np.log1p(np.exp(np.minimum(y, 20.0))) / 100.0)


def sigmoid100(z):
    y = 100.0 * z
    return np.where(y > 20.0, 1.0, 1.0 / (1.0 + np.exp(-np.minimum(y, 20.0))))


def sigmoid100_prime(z):
    s = sigmoid100(z)
    return np.where(100.0 * z > 20.0, 0.0, 100.0 * s * (1.0 - s))


# ------------------------------------------------------------------------------------------------ SDF value + normal
def sdf_forward(net, x, need_normal=True):
    """-> y[M, Nout], n[M, 3], cache     (idr.py:77-107; App. E forward)"""
    x = np.asarray(x, np.float64)
    d0 = 3 + 6 * net.multires
    h0 = pe(x, net.multires)
    a, A, Z = h0, [], []
    L = net.n_layers
    for l in range(L):
        if l in net.skip_layers:
            a = np.concatenate([a, h0], 1) / SQRT2
        A.append(a)
        z = a @ net.W[l].T + net.b[l]
        if l < L - 1:
            Z.append(z)
            a = softplus100(z)
    y = z
    cache = dict(x=x, h0=h0, A=A, Z=Z)
    if not need_normal:
        return y, None, cache
    M = x.shape[0]
    U = [None] * (L + 1)                       # U[l+1] multiplies sigma_l
    wl = net.W[L - 1][0]
    e = np.zeros((M, d0))
    if (L - 1) in net.skip_layers:             # a skip into the last Linear (idr.py:46-49,86): u_L splits like any skip layer's adjoint
        e = e + wl[-d0:] / SQRT2
        wl = wl[:-d0] / SQRT2
    U[L - 1] = np.broadcast_to(wl, (M, wl.shape[0]))
    for l in range(L - 2, -1, -1):
        s = sigmoid100(Z[l]) * U[l + 1]
        v = s @ net.W[l]
        if l in net.skip_layers:
            U[l] = v[:, :-d0] / SQRT2
            e = e + v[:, -d0:] / SQRT2
        else:
            U[l] = v
    g0 = U[0] + e
    n = _pe_jt(h0, g0, net.multires)
    cache.update(U=U, g0=g0)
    return y, n, cache


def _pe_jt(h0, g, multires):
    """J0^T g   (J0 = dPE/dx)"""
    n = g[:, :3].copy()
    for m in range(multires):
        f = 2.0 ** m
        s, c = h0[:, 3 + 6 * m:6 + 6 * m], h0[:, 6 + 6 * m:9 + 6 * m]
        n += f * (c * g[:, 3 + 6 * m:6 + 6 * m] - s * g[:, 6 + 6 * m:9 + 6 * m])
    return n


def _pe_j(h0, nb, multires):
    """J0 nbar"""
    out = [nb]
    for m in range(multires):
        f = 2.0 ** m
        s, c = h0[:, 3 + 6 * m:6 + 6 * m], h0[:, 6 + 6 * m:9 + 6 * m]
        out += [f * c * nb, -f * s * nb]
    return np.concatenate(out, 1)


def sdf_backward(net, cache, dy, dn=None, want_dx=True):
    """App. E backward.  dy[M, Nout], dn[M, 3] or None -> dW[l], db[l], dx"""
    L, d0 = net.n_layers, 3 + 6 * net.multires
    h0, A, Z = cache['h0'], cache['A'], cache['Z']
    dy = np.asarray(dy, np.float64)
    M = dy.shape[0]
    dW = [np.zeros_like(w) for w in net.W]
    db = [np.zeros_like(b) for b in net.b]
    ZB2 = [np.zeros_like(z) for z in Z]
    if dn is not None:
        dn = np.asarray(dn, np.float64)
        U = cache['U']
        gb0 = _pe_j(h0, dn, net.multires)
        ub = gb0
        for l in range(L - 1):
            vb = np.concatenate([ub, gb0], 1) / SQRT2 if l in net.skip_layers else ub
            sb = vb @ net.W[l].T
            sig = sigmoid100(Z[l])
            dW[l] += (sig * U[l + 1]).T @ vb
            ub = sig * sb
            ZB2[l] = U[l + 1] * sb * sigmoid100_prime(Z[l])
        vb = np.concatenate([ub, gb0], 1) / SQRT2 if (L - 1) in net.skip_layers else ub
        dW[L - 1][0] += vb.sum(0)
    dW[L - 1] += dy.T @ A[L - 1]
    db[L - 1] += dy.sum(0)
    hb = dy @ net.W[L - 1]
    h0b = np.zeros((M, d0))
    if (L - 1) in net.skip_layers:
        h0b += hb[:, -d0:] / SQRT2
        hb = hb[:, :-d0] / SQRT2
    for l in range(L - 2, -1, -1):
        zb = sigmoid100(Z[l]) * hb + ZB2[l]
        dW[l] += zb.T @ A[l]
        db[l] += zb.sum(0)
        ab = zb @ net.W[l]
        if l in net.skip_layers:
            hb = ab[:, :-d0] / SQRT2
            h0b += ab[:, -d0:] / SQRT2
        elif l == 0:
            h0b += ab
        else:
            hb = ab
    dx = None
    if want_dx:
        dx = _pe_jt(h0, h0b, net.multires)
        if dn is not None:
            g0 = cache['g0']
            for m in range(net.multires):
                f = 2.0 ** m
                s, c = h0[:, 3 + 6 * m:6 + 6 * m], h0[:, 6 + 6 * m:9 + 6 * m]
                dx += -(f * f) * (s * g0[:, 3 + 6 * m:6 + 6 * m] + c * g0[:, 6 + 6 * m:9 + 6 * m]) * dn
    return dW, db, dx


# ------------------------------------------------------------------------------------------------ rendering network
def render_forward(net, points, normals, view, feat, multires_view=4):
    """idr.py:145-167, mode 'idr'."""
    x = np.concatenate([np.asarray(points, np.float64), pe(view, multires_view), np.asarray(normals, np.float64),
                        np.asarray(feat, np.float64)], 1)
    A = []
    for l in range(net.n_layers):
        A.append(x)
        x = x @ net.W[l].T + net.b[l]
        if l < net.n_layers - 1:
            x = np.maximum(x, 0.0)
    rgb = np.tanh(x)
    return rgb, dict(A=A, rgb=rgb)


def render_backward(net, cache, drgb, multires_view=4):
    zb = np.asarray(drgb, np.float64) * (1.0 - cache['rgb'] ** 2)
    dW, db = [None] * net.n_layers, [None] * net.n_layers
    for l in range(net.n_layers - 1, -1, -1):
        dW[l] = zb.T @ cache['A'][l]
        db[l] = zb.sum(0)
        ab = zb @ net.W[l]
        if l > 0:
            zb = ab * (cache['A'][l] > 0)
    dv = 3 + 6 * multires_view
    return dW, db, ab[:, :3], ab[:, 3 + dv:6 + dv], ab[:, 6 + dv:]


# ------------------------------------------------------------------------------------------------ sample network
def sample_network(surface_output, surface_sdf_values, surface_points_grad, surface_dists, surface_cam_loc, surface_ray_dirs):
    """model/sample_network.py:10-20"""
    dot = (surface_points_grad * surface_ray_dirs).sum(1, keepdims=True)
    t = surface_dists - (surface_output - surface_sdf_values) / dot
    return surface_cam_loc + t * surface_ray_dirs


# ------------------------------------------------------------------------------------------------ feature consistency
def _project(pts_world, cam):
    """idx_world2cam + idx_cam2img (utils/my_utils.py:98-110): pts [m,3] world, cam[2,4,4] -> uv [m,2] (+ jacobian duv/dx [m,2,3])"""
    E, K = cam[0], cam[1][:3, :3]
    m = pts_world.shape[0]
    ph = np.concatenate([pts_world, np.ones((m, 1))], 1)
    c = ph @ E.T                                   # [m,4]
    c = c / (c[:, 3:4] + 1e-9)
    c3 = c[:, :3] / (c[:, 3:4] + 1e-9)
    i = c3 @ K.T
    uv = i[:, :2] / (i[:, 2:3] + 1e-9)
    return uv


def _bilinear(fmap, gx, gy):
    """F.grid_sample(bilinear, zeros, align_corners=False) on one CHW map at normalised coords -> [C,m], d/dgx, d/dgy"""
    C, H, W = fmap.shape
    ix = ((gx + 1.0) * W - 1.0) / 2.0
    iy = ((gy + 1.0) * H - 1.0) / 2.0
    x0, y0 = np.floor(ix), np.floor(iy)
    fx, fy = ix - x0, iy - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = fmap[:, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return v * ok[None]
    v00, v01, v10, v11 = tap(y0, x0), tap(y0, x0 + 1), tap(y0 + 1, x0), tap(y0 + 1, x0 + 1)
    val = v00 * (1 - fx) * (1 - fy) + v01 * fx * (1 - fy) + v10 * (1 - fx) * fy + v11 * fx * fy
    dix = (v01 - v00) * (1 - fy) + (v11 - v10) * fy
    diy = (v10 - v00) * (1 - fx) + (v11 - v01) * fx
    return val, dix * (W / 2.0), diy * (H / 2.0)


def _project_jac(pts_world, cam):
    """uv of _project and its jacobian d uv / d x [m,2,3] for an affine extrinsic (last row 0 0 0 1; the 1e-9 guards of the reference enter at 1e-9 relative)"""
    E, K = np.asarray(cam[0], np.float64), np.asarray(cam[1], np.float64)[:3, :3]
    c3 = pts_world @ E[:3, :3].T + E[:3, 3]
    i = c3 @ K.T
    iz = i[:, 2:3] + 1e-9
    uv = i[:, :2] / iz
    dK = (K[None, :2, :] - uv[:, :, None] * K[None, 2:3, :]) / iz[:, :, None]          # d uv / d c3  [m,2,3]
    return uv, dK @ E[:3, :3]


def feat_corr_loss(diff_surf_pts, hit_counts, feat, cam, feat_src, src_cams, size, center, with_grad=False, eps=1e-6):
    """IDRLoss.get_feat_loss_corr (model/loss.py:115-165).  diff_surf_pts [N,3] (hit points, view-major), hit_counts[B].
    with_grad=True: also the ANALYTIC gradient w.r.t. the points (projection jacobian x bilinear-tap derivatives x the derivative of the normalised
    correlation; what the reference's autograd and csrc/loss_kernels.hip::k_feat_corr compute); with_grad='fd': central differences of the (piecewise smooth)
    loss instead -- O(N) evaluations, the cross-check of the analytic form on small inputs (tests/test_oracle_np_golden.py)."""
    pts = np.asarray(diff_surf_pts, np.float64)
    ctr = np.asarray(center, np.float64).reshape(1, 3)
    nb = len(hit_counts)

    def total(p, grad=None):
        losses, start = [], 0
        for b, cnt in enumerate(hit_counts):
            cnt = int(cnt)
            if cnt == 0:
                losses.append(0.0)
                continue
            pw = p[start:start + cnt] / 2.0 * float(size) + ctr
            cams = [cam[b]] + [src_cams[b][v] for v in range(src_cams.shape[1])]
            fmaps = [feat[b]] + [feat_src[b][v] for v in range(feat_src.shape[1])]
            vals, inr, dvals = [], [], []
            for cm, fm in zip(cams, fmaps):
                if grad is None:
                    uv, J = _project(pw, np.asarray(cm, np.float64)), None
                else:
                    uv, J = _project_jac(pw, cm)
                uv = uv / 2.0
                H, W = fm.shape[1:]
                gx_raw, gy_raw = uv[:, 0] / W * 2 - 1, uv[:, 1] / H * 2 - 1
                gx = np.clip(gx_raw, -1.1, 1.1)
                gy = np.clip(gy_raw, -1.1, 1.1)
                inr.append((gx <= 1) & (gx >= -1) & (gy <= 1) & (gy >= -1))
                val, dgx, dgy = _bilinear(np.asarray(fm, np.float64), gx, gy)
                vals.append(val)
                if grad is not None:
                    # d val / d p [C,m,3]: gx = u / W - 1 with u = uv_proj[0] (the / 2 and * 2 cancel), zero where the clip is active; d pw / d p = size / 2
                    ux = (np.abs(gx_raw) < 1.1)[None, :, None] * dgx[:, :, None] * J[None, :, 0, :] / W
                    uy = (np.abs(gy_raw) < 1.1)[None, :, None] * dgy[:, :, None] * J[None, :, 1, :] / H
                    dvals.append((ux + uy) * (float(size) / 2.0))
            n0r = np.linalg.norm(vals[0], axis=0)
            n0 = np.maximum(n0r, 1e-9)
            acc = 0.0
            V = len(cams) - 1
            for v in range(1, V + 1):
                nvr = np.linalg.norm(vals[v], axis=0)
                nv = np.maximum(nvr, 1e-9)
                corr = (vals[0] * vals[v]).sum(0) / n0 / nv
                cl = np.abs(1 - corr)
                m = (inr[0] & inr[v]) * (cl < 0.5)
                acc += (cl * m).sum()
                if grad is not None:
                    # d corr = <da, b> / (n0 nv) - corr <a, da> / n0^2 (norm above its floor) + the same with a <-> b
                    a_, b_ = vals[0], vals[v]
                    dc_da = b_ / (n0 * nv) - (n0r > 1e-9) * corr * a_ / (n0 * n0)
                    dc_db = a_ / (n0 * nv) - (nvr > 1e-9) * corr * b_ / (nv * nv)
                    dcorr = np.einsum('cm,cmk->mk', dc_da, dvals[0]) + np.einsum('cm,cmk->mk', dc_db, dvals[v])
                    grad[start:start + cnt] += (-np.sign(1 - corr) * m)[:, None] * dcorr / (V * cnt) / nb
            start += cnt
            losses.append(acc / (V * cnt))
        return sum(losses) / len(losses)
    if with_grad == 'fd':
        loss = total(pts)
        g = np.zeros_like(pts)
        for i in range(pts.shape[0]):
            for c in range(3):
                p1, p2 = pts.copy(), pts.copy()
                p1[i, c] += eps
                p2[i, c] -= eps
                g[i, c] = (total(p1) - total(p2)) / (2 * eps)
        return loss, g
    if not with_grad:
        return total(pts)
    g = np.zeros_like(pts)
    loss = total(pts, g)
    return loss, g


# ------------------------------------------------------------------------------------------------ depth carving + loss terms
def carving_t2(points_world, depths, cams, out_thresh_perc=1 / 8, use_invalid=False):
    """utils/my_utils.py:269-331 for n = 1: points [m,3] world, depths [v,h,w], cams [v,2,4,4] -> dist[m], inside[m], valid[m].
    use_invalid: carving_t (my_utils.py:204-266, conf.use_invalid): an in-range view without a depth is half an 'outside' vote; third result = in-range mask."""
    m, v = points_world.shape[0], depths.shape[0]
    MAXF = 1e30 / v
    tot_in, tot_valid, tot_inside = np.zeros(m), np.zeros(m), np.zeros(m)
    pos_min, neg_max = np.full(m, np.inf), np.full(m, -np.inf)
    for i in range(v):
        cam = np.asarray(cams[i], np.float64)
        E, K = cam[0], cam[1][:3, :3]
        ph = np.concatenate([points_world, np.ones((m, 1))], 1) @ E.T
        ph = ph / (ph[:, 3:4] + 1e-9)
        pdepth = ph[:, 2]
        c3 = ph[:, :3] / (ph[:, 3:4] + 1e-9)
        im = c3 @ K.T
        uv = im[:, :2] / (im[:, 2:3] + 1e-9)
        h, w = depths[i].shape
        gx = np.clip(uv[:, 0] / w * 2 - 1, -1.1, 1.1)
        gy = np.clip(uv[:, 1] / h * 2 - 1, -1.1, 1.1)
        in_range = (gx <= 1) & (gx >= -1) & (gy <= 1) & (gy >= -1)
        ix = np.round(((gx + 1) * w - 1) / 2).astype(np.int64)        # nearest, align_corners=False (half-to-even like nearbyint)
        iy = np.round(((gy + 1) * h - 1) / 2).astype(np.int64)
        ok = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
        gd = np.where(ok, depths[i][np.clip(iy, 0, h - 1), np.clip(ix, 0, w - 1)], 0.0)
        valid = (gd > 0) & in_range
        inside = (pdepth > gd * 0.99) & valid
        outside = valid ^ inside
        dist = (pdepth - gd) * valid
        tot_in += in_range
        tot_valid += valid
        tot_inside += inside
        pos_min = np.minimum(pos_min, np.where(inside, dist, MAXF))
        neg_max = np.maximum(neg_max, np.where(outside, dist, -MAXF))

    def agg(res, sign):
        validm = np.abs(res) < MAXF * .99
        num = validm.astype(np.float64)
        ret = (res * validm) / (num + 1e-9)
        return ret * (num > 0.5) + MAXF * sign * (num < 0.5)
    dpos, dneg = agg(pos_min, 1), agg(neg_max, -1)
    if use_invalid:
        outside_perc = ((tot_valid - tot_inside) + (tot_in - tot_valid) * 0.5) / (tot_in + 1e-9)
        scene_valid = tot_in > 0
    else:
        outside_perc = (tot_valid - tot_inside) / (tot_valid + 1e-9)
        scene_valid = tot_valid > 0
    scene_outside = (outside_perc > out_thresh_perc) & scene_valid
    scene_inside = scene_valid ^ scene_outside
    return dpos * scene_inside + dneg * scene_outside, scene_inside, scene_valid


def depth_loss(eik_points, eik_output, depths, depth_cams, size, center, far_thresh=0.25, far_att=1, near_thresh=0.1, near_att=1, smooth=None, use_invalid=False):
    """model/loss.py:37-63 (use_invalid=False; smooth: loss.py:57-58, SmoothL1(eo / s, -dist_r / s) * s with beta = 1).  eik_points [M,3] normalised, eik_output [M]; depths [B,1,1,h,w]."""
    pw = np.asarray(eik_points, np.float64) / 2 * float(size) + np.asarray(center, np.float64).reshape(1, 3)
    dist, _, in_range = carving_t2(pw, np.asarray(depths, np.float64)[:, 0, 0], np.asarray(depth_cams, np.float64)[:, 0], use_invalid=use_invalid)
    dist_r = np.clip(dist / float(size) * 2 + (-1.25) * (~in_range), -1.25, 1.25)
    far = np.abs(dist_r) > far_thresh
    near = np.abs(dist_r) < near_thresh
    w = (far * far_att + ~far) * (near * near_att + ~near) * in_range
    df = np.asarray(eik_output, np.float64) + dist_r
    el = np.abs(df)
    if smooth is not None:
        x = df / smooth
        el = np.where(np.abs(x) < 1, 0.5 * x * x, np.abs(x) - 0.5) * smooth
    return (el * w).mean(), dist_r, w


def dsurf_unproject(depths, depth_cams, size, center):
    """Phase-0 depth-surface points (reference idr.py:234-238 with my_utils.py:71-95): every depth pixel (b, y, x) unprojected through
    the inverse intrinsics / extrinsics of its depth camera and normalised with (p - center) / size * 2.
    depths [N,H,W], depth_cams [N,2,4,4] -> (points [N,H,W,3] float64, valid [N,H,W] = depth > 0)."""
    depths = np.asarray(depths, np.float64)
    cams = np.asarray(depth_cams, np.float64)
    N, H, W = depths.shape
    xs, ys = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)               # get_pixel_grids
    pix = np.stack([xs, ys, np.ones_like(xs)], -1)                             # H W 3
    out = np.zeros((N, H, W, 3))
    for n in range(N):
        kinv = np.linalg.inv(cams[n, 1, :3, :3])
        einv = np.linalg.inv(cams[n, 0])
        ic = pix @ kinv.T                                                      # idx_img2cam
        ic = ic / (ic[..., 2:3] + 1e-9) * depths[n][..., None]
        hom = np.concatenate([ic, np.ones_like(ic[..., :1])], -1)
        w = hom @ einv.T                                                       # idx_cam2world
        w = w / (w[..., 3:4] + 1e-9)
        out[n] = (w[..., :3] - np.asarray(center, np.float64).reshape(-1)[:3]) / float(np.asarray(size).reshape(-1)[0]) * 2
    return out, depths > 0
