"""torch.autograd.Function wrappers around the HIP passes (mvsdf_amd.ops).  All math happens in libmvsdf_hip.so;
this file only routes tensors and gradients so that `loss.backward()` (idr_train.py:287) drives the hand-written
backward kernels instead of autograd's op-by-op double backward."""
import torch

from . import ops


class grad_sink:
    """Context manager that lets _FoldNet.backward ADD the parameter gradients straight into the persistent flat `.grad` buffers (views
    set up by optim.FlatAdam / parallel.FlatGradBucket) instead of returning them to autograd -- one launch instead of ~3 AccumulateGrad
    launches per layer.  Opt-in because it bypasses autograd's bookkeeping: under it `torch.autograd.grad(loss, params)`,
    `backward(inputs=...)` and parameter hooks would see no gradients.  FlatAdam.backward(loss) / FlatGradBucket.backward(loss) wrap
    `loss.backward()` in it; a plain `loss.backward()` takes the ordinary autograd route (same numbers)."""
    depth = 0

    def __enter__(self):
        grad_sink.depth += 1
        return self

    def __exit__(self, *exc):
        grad_sink.depth -= 1
        return False


def mark_sink_written(params):
    """A kernel ADDED into the parameters' .grad buffers through raw pointers: move their version counter (the views of one flat buffer share it) so that whoever
    tracks the buffer -- optim.FlatAdam.zero_grad skips its memset while the zeros its last step left are untouched -- sees the write."""
    for p in params:
        if p is not None and p.grad is not None:
            torch.autograd.graph.increment_version(p.grad)
            return


class _Fold(torch.autograd.Function):
    """weight_norm (idr.py:70-71): (weight_v, weight_g) -> folded W; MFMA packs ride along on `holder`."""

    @staticmethod
    def forward(ctx, v, g, holder):
        w, wp, wpT = ops.fold_pack(v.detach(), g.detach())
        holder.w, holder.wp, holder.wpT = w, wp, wpT
        ctx.save_for_backward(v, g)
        return w

    @staticmethod
    def backward(ctx, dW):
        v, g = ctx.saved_tensors
        dv, dg = ops.fold_backward(v.detach(), g.detach(), dW.contiguous())
        return dv, dg.view_as(g), None


class _FoldNet(torch.autograd.Function):
    """weight_norm of a whole network in one C call (one fold launch + one pack launch): (*weight_v, *weight_g, *bias) -> (*W, *bias).
    Biases pass through so that the backward sees every gradient of the network: inside a `grad_sink()` context, when the parameters
    carry the `_mv_grad_sink` mark (set by optim.FlatAdam / parallel.FlatGradBucket: their .grad buffers are persistent views of one flat
    buffer) the single backward launch ADDS dv, dg and db straight into those buffers and returns no gradients -- the work of ~3
    AccumulateGrad launches per layer."""

    @staticmethod
    def forward(ctx, holders, *vgb):
        n = len(vgb) // 3
        vs, gs = [t.detach() for t in vgb[:n]], [t.detach() if t is not None else None for t in vgb[n:2 * n]]   # g None: weight_norm=False, w = v
        ws, wps, wpTs = ops.fold_pack_net(vs, gs)
        for L, w, wp, wpT in zip(holders, ws, wps, wpTs):
            L.w, L.wp, L.wpT = w, wp, wpT
        ctx.params = vgb
        ctx.n = n
        return tuple(ws) + tuple(b.view_as(b) for b in vgb[2 * n:])

    @staticmethod
    def backward(ctx, *grads):
        n, vgb = ctx.n, ctx.params
        vs, gs, bs = vgb[:n], vgb[n:2 * n], vgb[2 * n:]
        dWs = [d.contiguous() if d is not None else torch.zeros_like(v) for d, v in zip(grads[:n], vs)]
        dbs = [d.contiguous() if d is not None else torch.zeros_like(b) for d, b in zip(grads[n:], bs)]
        vd, gd = [t.detach() for t in vs], [t.detach() if t is not None else None for t in gs]
        direct = grad_sink.depth > 0 and all(getattr(p, '_mv_grad_sink', False) and p.grad is not None and p.grad.is_contiguous() and p.requires_grad
                                             for p in vgb if p is not None)
        if direct:
            ops.fold_backward_net(vd, gd, dWs, dbs, sinks=([p.grad for p in vs], [p.grad if p is not None else None for p in gs], [p.grad for p in bs]))
            mark_sink_written(vs)
            return (None,) * (1 + 3 * n)
        dvs, dgs = ops.fold_backward_net(vd, gd, dWs)
        return (None,) + tuple(dvs) + tuple(dg.view_as(g) if g is not None else None for dg, g in zip(dgs, gs)) + tuple(dbs)


class _FoldNetFlat(torch.autograd.Function):
    """_FoldNet for the training step: the folded weights of all networks are ONE tensor [W | b per network] (ops.FoldPlan), so the step's
    autograd graph has one edge into the fold instead of one per layer tensor, its backward receives one flat gradient (the backward kernels
    already emit dW_cat / db_cat per network) and all per-layer bookkeeping is pointer arithmetic on cached ctypes arrays."""

    @staticmethod
    def forward(ctx, plan, holders, *vgb):
        n = plan.n
        flat, packs = ops.fold_pack_net_flat(plan, vgb[:n], vgb[n:2 * n], vgb[2 * n:], holders)
        for L in holders:
            L.wp16 = None
            L.keep = packs                                       # the layers carry raw pointers into `packs` / `flat`: keep both alive with them
        ctx.plan, ctx.params = plan, vgb
        return flat

    @staticmethod
    def backward(ctx, dflat):
        plan, vgb = ctx.plan, ctx.params
        n = plan.n
        vs, gs, bs = vgb[:n], vgb[n:2 * n], vgb[2 * n:]
        sink = grad_sink.depth > 0 and all(getattr(p, '_mv_grad_sink', False) and p.requires_grad for p in vgb if p is not None)
        res = ops.fold_backward_net_flat(plan, vs, gs, bs, dflat.contiguous(), sink)
        if res is None:
            mark_sink_written(vs)
            return (None,) * (2 + 3 * n)
        dvs, dgs, dbs = res
        return (None, None) + tuple(dvs) + tuple(dg.view_as(g) if g is not None else None for dg, g in zip(dgs, gs)) + tuple(dbs)


def fold_networks_flat(specs, cache):
    """fold_networks with the folded parameters of all networks in one flat tensor.  cache: a dict owned by the model (keeps the FoldPlan).
    -> (flat, plan, [PackedNet per spec])."""
    shapes, cuts, vs_all, gs_all, bs_all = [], [], [], [], []
    for vs, gs, bs, _, _ in specs:
        shapes += [tuple(v.shape) for v in vs]
        cuts.append(len(shapes))
        vs_all += list(vs); gs_all += list(gs); bs_all += list(bs)
    plan = cache.get('plan')
    if plan is None or plan.shapes != shapes:
        plan = cache['plan'] = ops.FoldPlan(shapes, cuts)
    layers = []
    for v, b in zip(vs_all, bs_all):
        L = ops.PackedLayer()
        L.bias = b.detach()
        L.N, L.K = v.shape
        layers.append(L)
    flat = _FoldNetFlat.apply(plan, layers, *vs_all, *gs_all, *bs_all)
    nets, lo = [], 0
    for (vs, gs, bs, skip_layer, multires), hi in zip(specs, cuts):
        nets.append(ops.maybe_pack_x3_chain(ops.PackedNet(layers[lo:hi], skip_layer, multires)))
        lo = hi
    for net in nets:
        net._keep = flat                                         # raw pointers into `flat` / its packs: keep them alive with the net
    return flat, plan, nets


def fold_networks(specs):
    """Several networks folded by ONE autograd node (one fold launch + one pack launch forward, one launch backward).
    specs: list of (vs, gs, bs, skip_layer, multires) -> list of (PackedNet, [w linked to autograd], [biases linked to autograd])."""
    layers, vs_all, gs_all, bs_all, cuts = [], [], [], [], []
    for vs, gs, bs, _, _ in specs:
        for v, b in zip(vs, bs):
            L = ops.PackedLayer()
            L.bias = b.detach()
            L.N, L.K = v.shape
            L.wp16 = None
            layers.append(L)
        vs_all += list(vs); gs_all += list(gs); bs_all += list(bs)
        cuts.append(len(layers))
    n = len(layers)
    out = _FoldNet.apply(layers, *vs_all, *gs_all, *bs_all)
    res, lo = [], 0
    for (vs, gs, bs, skip_layer, multires), hi in zip(specs, cuts):
        res.append((ops.maybe_pack_x3_chain(ops.PackedNet(layers[lo:hi], skip_layer, multires)), list(out[lo:hi]), list(out[n + lo:n + hi])))
        lo = hi
    return res


def fold_network(vs, gs, bs, skip_layer, multires):
    """-> (PackedNet, [w tensors linked to autograd], [biases linked to autograd])"""
    layers = []
    for v, b in zip(vs, bs):
        L = ops.PackedLayer()
        L.bias = b.detach()
        L.N, L.K = v.shape
        L.wp16 = None
        layers.append(L)
    n = len(vs)
    out = _FoldNet.apply(layers, *vs, *gs, *bs)
    return ops.maybe_pack_x3_chain(ops.PackedNet(layers, skip_layer, multires)), list(out[:n]), list(out[n:])


class SharedSdfEval:
    """What one fused evaluation leaves behind for the backward passes and for re-use at the same points."""
    __slots__ = ('net', 'x', 'M', 'Mg', 'n_active', 'saved', 'y', 'n', 'stash')


class _SdfValueNormal(torch.autograd.Function):
    """x[M,3] -> y[M, 1+1+F], n[Mg,3]   (ImplicitNetwork.forward + .gradient, idr.py:77-107).
    Rows >= n_active are known not to receive gradients (non-hit rays): the backward skips them."""

    @staticmethod
    def forward(ctx, x, shared, *wb):
        y, n, saved = ops.sdf_forward(shared.net, x.detach(), shared.Mg)
        shared.x, shared.saved, shared.y, shared.n = x.detach(), saved, y, n
        ctx.shared = shared
        ctx.nl = len(wb) // 2
        ctx.x_needs_grad = x.requires_grad
        return y, n

    @staticmethod
    def backward(ctx, dy, dn):
        sh = ctx.shared
        Mb = sh.n_active
        if Mb == 0:
            return (None, None) + tuple(torch.zeros_like(L.w) for L in sh.net.layers) + tuple(torch.zeros_like(L.bias) for L in sh.net.layers)
        Nout = sh.net.layers[-1].N
        dyb = dy[:Mb].contiguous() if dy is not None else torch.zeros(Mb, Nout, device=sh.x.device)
        if sh.stash is not None:                         # upstream grads of the re-use at the surface points (same rows, same weights):
            r0, n0, sdy, sdn = sh.stash                  # by linearity their weight gradients are computed here, once
            sh.stash = None
            if dyb.data_ptr() == (dy.data_ptr() if dy is not None else 0):
                dyb = dyb.clone()
            dyb[r0:r0 + n0] += sdy
            if sdn is not None:
                if dn is None:
                    dn = torch.zeros(sh.Mg, 3, device=sh.x.device)
                else:
                    dn = dn.clone()
                dn[r0:r0 + n0] += sdn
        dnb = None
        if dn is not None and sh.Mg > 0:
            mg = min(Mb, sh.Mg)
            if mg < Mb:                                  # normals cover a shorter prefix: split by linearity
                dWa, dba, dxa = ops.sdf_backward(sh.net, sh.x, sh.M, sh.Mg, mg, dyb[:mg].contiguous(), dn[:mg].contiguous(), sh.saved,
                                                 ctx.x_needs_grad)
                rest = dyb.clone()
                rest[:mg] = 0
                dWb, dbb, dxb = ops.sdf_backward(sh.net, sh.x, sh.M, sh.Mg, Mb, rest, None, sh.saved, ctx.x_needs_grad)
                dWs = [a + b for a, b in zip(dWa, dWb)]
                dbs = [a + b for a, b in zip(dba, dbb)]
                dx = None
                if ctx.x_needs_grad:
                    dx = torch.zeros_like(sh.x)
                    dx[:Mb] = dxb
                    dx[:mg] += dxa
                return (dx, None) + tuple(dWs) + tuple(dbs)
            dnb = dn[:Mb].contiguous()
        dWs, dbs, dxb = ops.sdf_backward(sh.net, sh.x, sh.M, sh.Mg, Mb, dyb, dnb, sh.saved, ctx.x_needs_grad)
        dx = None
        if ctx.x_needs_grad:
            dx = torch.zeros_like(sh.x)
            dx[:Mb] = dxb
        return (dx, None) + tuple(dWs) + tuple(dbs)


def sdf_value_normal(net, ws, bs, x, Mg, n_active=None):
    sh = SharedSdfEval()
    sh.net, sh.M, sh.Mg = net, x.shape[0], Mg
    sh.n_active = x.shape[0] if n_active is None else n_active
    sh.stash = None
    y, n = _SdfValueNormal.apply(x, sh, *ws, *bs)
    return y, n, sh


class _SdfReuse(torch.autograd.Function):
    """Value + normal at points that are numerically rows [row0, row0 + N) of an earlier evaluation (the differentiable
    surface points x(theta) equal the traced points: sample_network.py:14 with f - f0 == 0).  Forward re-uses the stored
    outputs; backward runs the full first/second-order backward on those rows, including d/dx (idr.py:325-326)."""

    @staticmethod
    def forward(ctx, pts, shared, row0, N, defer_dw, *wb):
        ctx.shared, ctx.row0, ctx.N, ctx.defer_dw = shared, row0, N, defer_dw
        return shared.y[row0:row0 + N].clone(), shared.n[row0:row0 + N].clone()

    @staticmethod
    def backward(ctx, dy, dn):
        sh, N, r0 = ctx.shared, ctx.N, ctx.row0
        Nout = sh.net.layers[-1].N
        dyb = dy.contiguous() if dy is not None else torch.zeros(N, Nout, device=sh.x.device)
        dnb = dn.contiguous() if dn is not None else None
        if ctx.defer_dw:
            # only d/dx is computed here; (dy, dn) are stashed and folded into the main evaluation's backward, which autograd runs
            # later (its outputs feed sample_network -> these points), so the weight gradients of both uses come from ONE pass
            _, _, dx = ops.sdf_backward(sh.net, sh.x, sh.M, sh.Mg, N, dyb, dnb, sh.saved, True, want_dw=False, row0=r0)
            sh.stash = (r0, N, dyb, dnb)
            return (dx, None, None, None, None) + (None,) * (2 * len(sh.net.layers))
        dWs, dbs, dx = ops.sdf_backward(sh.net, sh.x, sh.M, sh.Mg, N, dyb, dnb, sh.saved, True, row0=r0)
        return (dx, None, None, None, None) + tuple(dWs) + tuple(dbs)


def sdf_reuse(shared, ws, bs, pts, N, defer_dw=False, row0=0):
    """defer_dw: True only when `pts` is a differentiable function of the main evaluation's outputs (training: sample_network),
    so that the main backward is guaranteed to run after this one and can take over the weight gradients."""
    return _SdfReuse.apply(pts, shared, row0, N, defer_dw, *ws, *bs)


def render_offsets(vspec):
    """Column layout of the rendering net's input for a view spec (multires_view | 0x100 if mode 'no_view_dir' | 0x200 if 'no_normal'):
    -> (first normal column or -1, first feature column)."""
    dv = 0 if vspec & 0x100 else 3 + 6 * (vspec & 0xff)
    dn = 0 if vspec & 0x200 else 3
    return (3 + dv if dn else -1), 3 + dv + dn


class _Render(torch.autograd.Function):
    """RenderingNetwork.forward (idr.py:145-167); multires_view carries the mode bits (render_offsets)."""

    @staticmethod
    def forward(ctx, points, normals, view, feat, net, multires_view, *wb):
        N = points.shape[0]
        fd = feat.detach()
        if fd.stride(1) != 1:
            fd = fd.contiguous()
        rgb, saved = ops.render_forward(net, points.detach(), view.detach(), normals.detach(), fd, multires_view)
        ctx.net, ctx.saved, ctx.N, ctx.mv = net, saved, N, multires_view
        ctx.needs = (points.requires_grad, normals.requires_grad, feat.requires_grad)
        return rgb

    @staticmethod
    def backward(ctx, drgb):
        dWs, dbs, din = ops.render_backward(ctx.net, ctx.N, drgb.contiguous(), ctx.saved)
        nrm0, feat0 = render_offsets(ctx.mv)
        dp = din[:, 0:3] if ctx.needs[0] else None
        dn = (din[:, nrm0:nrm0 + 3] if nrm0 >= 0 else torch.zeros_like(din[:, :3])) if ctx.needs[1] else None
        df = din[:, feat0:] if ctx.needs[2] else None
        return (dp, dn, None, df, None, None) + tuple(dWs) + tuple(dbs)


def render(net, ws, bs, points, normals, view, feat, multires_view):
    return _Render.apply(points, normals, view, feat, net, multires_view, *ws, *bs)


class _FeatCorr(torch.autograd.Function):
    """IDRLoss.get_feat_loss_corr (loss.py:115-165): loss and d/d(points) come out of one kernel launch."""

    @staticmethod
    def forward(ctx, pts, view_start, feat, feat_src, cam, src_cams, size, center):
        loss_pp, dpts = ops.feat_corr(pts.detach(), view_start, feat, feat_src, cam, src_cams, size, center)
        ctx.save_for_backward(dpts)
        return loss_pp.sum()

    @staticmethod
    def backward(ctx, g):
        (dpts,) = ctx.saved_tensors
        return dpts * g, None, None, None, None, None, None, None


def feat_corr_loss(pts, view_start, feat, feat_src, cam, src_cams, size, center):
    return _FeatCorr.apply(pts, view_start, feat, feat_src, cam, src_cams, size, center)


class _FeatCorrPP(torch.autograd.Function):
    """Per-point terms of the feature-consistency loss (their sum is the loss); backward scales the stored analytic gradient."""

    @staticmethod
    def forward(ctx, pts, view_start, feat, feat_src, cam, src_cams, size, center):
        loss_pp, dpts = ops.feat_corr(pts.detach(), view_start, feat, feat_src, cam, src_cams, size, center)
        ctx.save_for_backward(dpts)
        return loss_pp

    @staticmethod
    def backward(ctx, g):
        (dpts,) = ctx.saved_tensors
        return dpts * g.unsqueeze(-1), None, None, None, None, None, None, None


class _LossTerms(torch.autograd.Function):
    """rgb L1 + eikonal + depth L1 + surface BCE + feature sum + weighted total in one launch (loss.py:176-219).
    Returns the six scalars (loss, rgb, eikonal, depth, feat, surf) as separate 0-d tensors; backward = the stored unit gradients times
    (d/d loss * weight + d/d term), one launch."""

    @staticmethod
    def forward(ctx, rgb, grad_theta, eik_out, surf, feat_pp, rgb_gt, rgb_mask, dist_r, dweight, n_pos, weights, surf_on, feat_on, inv_counts=None):
        out, d_rgb, d_grad, d_eo, d_sf = ops.loss_terms(rgb.detach(), rgb_gt, rgb_mask, grad_theta.detach() if grad_theta is not None else None,
                                                        eik_out.detach(), dist_r, dweight, surf.detach() if surf is not None else None, n_pos,
                                                        feat_pp.detach() if feat_pp is not None else None, weights, surf_on, feat_on, inv_counts)
        ctx.saved = (d_rgb, d_grad, d_eo, d_sf)
        ctx.weights, ctx.shapes = weights, (eik_out.shape, feat_pp.shape if feat_pp is not None else None)
        ctx.on = (surf_on, feat_on)
        ctx.set_materialize_grads(False)                         # unused scalars arrive as None, not as zero tensors
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        d_rgb, d_grad, d_eo, d_sf = ctx.saved
        use_sf = d_sf is not None and ctx.on[0]
        (g_rgb, g_grad, g_eo, g_sf), c_feat = ops.loss_scale(gs, ctx.weights, [d_rgb, d_grad, d_eo, d_sf if use_sf else None])   # one launch
        g_fp = c_feat.expand(ctx.shapes[1]) if (ctx.shapes[1] is not None and ctx.on[1]) else None
        return g_rgb, g_grad, g_eo.view(ctx.shapes[0]), g_sf, g_fp, None, None, None, None, None, None, None, None, None


def feat_corr_terms(pts, view_start, feat, feat_src, cam, src_cams, size, center):
    return _FeatCorrPP.apply(pts, view_start, feat, feat_src, cam, src_cams, size, center)


def loss_terms(rgb, grad_theta, eik_out, surf, feat_pp, rgb_gt, rgb_mask, dist_r, dweight, n_pos, weights, surf_on, feat_on, inv_counts=None):
    return _LossTerms.apply(rgb, grad_theta, eik_out, surf, feat_pp, rgb_gt, rgb_mask, dist_r, dweight, n_pos, weights, surf_on, feat_on, inv_counts)


class StepState:
    """Everything the fused training step needs besides the (folded) parameters."""
    __slots__ = ('net', 'rnet', 'x_eval', 'y_eval', 'n_eval', 'saved', 'R', 'E', 'N', 'n_true', 'n_eik', 'n_ds', 'inv', 'perm', 'true_rows',
                 'view_sorted', 'counts_dev', 'wait_counts', 'd_mask', 'e_mask', 'detach_geo', 'multires_view', 'rsaved', 'sdf_output',
                 'points_hom', 'plan')


class _IdrStep(torch.autograd.Function):
    """Post-trace half of IDRNetwork.forward in training mode as ONE autograd node (idr.py:202-304): from the fused value + normal
    evaluation `state.y_eval / n_eval` (rows [samples | rays, hit first]) it produces diff_surf_pts, rgb_values, grad_theta,
    eikonal_output, surf_indicator_output; backward chains the rendering-net backward, the input adjoint at the surface points,
    SampleNetwork's scalar and a single first/second-order SDF backward -- no autograd glue in between.

    The forward enqueues its kernels BEFORE the host knows the hit count N: the rendering net runs on every sorted ray row and the
    gather kernel reads N from device memory into worst-case sized outputs.  Only then does the host wait for the counts
    (`state.wait_counts()`) and narrow the outputs to their N-dependent shapes (views, no kernels)."""

    @staticmethod
    def forward(ctx, st, *params):
        R, E = st.R, st.E
        rgb_sorted, st.rsaved = ops.render_forward(st.rnet, st.x_eval[E:], st.view_sorted, st.n_eval[E:], st.y_eval[E:, 2:], st.multires_view)
        rgb_values, sdf_output, diff_pts, eik_out, hom, gth, surf = ops.step_outputs(
            R, st.n_eik, st.n_ds, st.counts_dev, st.x_eval, st.y_eval, st.n_eval, st.inv, st.true_rows, rgb_sorted, st.d_mask, st.e_mask)
        N, n_true = st.wait_counts()                             # the one host wait of the training forward
        st.N, st.n_true = N, n_true
        sizes = (N, st.n_eik, st.n_ds, st.n_ds)
        nd = sum(c for g, c in enumerate(sizes) if st.d_mask >> g & 1)
        ne = sum(c for g, c in enumerate(sizes) if st.e_mask >> g & 1)
        st.sdf_output, st.points_hom = sdf_output, hom[:nd].view(1, nd, 4, 1)
        ctx.st = st
        return diff_pts[:N], rgb_values, gth[:ne], eik_out[:nd].view(1, nd), surf[:n_true + st.n_eik]

    @staticmethod
    def backward(ctx, d_diff, d_rgbv, d_gth, d_eo, d_si):
        st = ctx.st
        R, E, N = st.R, st.E, st.N
        dev = st.x_eval.device
        net, rnet = st.net, st.rnet
        M, Mb, Nout = R + E, E + N, net.layers[-1].N
        dy = torch.empty(Mb, Nout, dtype=torch.float32, device=dev)
        dn = torch.empty(Mb, 3, dtype=torch.float32, device=dev)
        dWr = dbr = din = dx = None
        use_geo = not st.detach_geo                                               # idr.py:329-336: features always carry the rgb gradient
        nrm0, feat0 = render_offsets(st.multires_view)
        plan = getattr(st, 'plan', None)                                          # flat mode: ONE gradient tensor [dW | db per network]
        out_s = out_r = dflat = None
        if plan is not None:
            dflat = torch.empty(plan.total, dtype=torch.float32, device=dev)
            (w0, b0, e0), (w1, b1, e1) = plan.seg
            out_s, out_r = (dflat[w0:b0], dflat[b0:e0]), (dflat[w1:b1], dflat[b1:e1])
        if N > 0 and d_rgbv is not None:
            dWr, dbr, din = ops.render_backward(rnet, N, d_rgbv[st.perm[:N]], st.rsaved, n_ctx=R, out=out_r)
        elif plan is not None:
            dflat[w1:e1].zero_()
        common = (st.n_eik, st.n_ds, N, Nout, st.n_true, din, feat0, nrm0, use_geo)
        ops.step_backward_inputs(0, *common, None, None, st.view_sorted, st.n_eval, st.true_rows, None, None, None, st.d_mask, st.e_mask, dy, dn)
        dWs = None
        if din is not None and N > 0:
            # The adjoint of the surface points needs ONE input-adjoint pass over the hit rows with the rendering net's upstream alone
            # (features + normals, rows [E, E+N)); SampleNetwork's scalar fbar = -xbar.v / n.v (App. E.6) then enters output column 0 of
            # the same rows.  By linearity the full backward is (A) a pass with every upstream EXCEPT fbar, which does not depend on the
            # input-adjoint pass and shares a launch with it, plus (B) a short first-order delta pass for fbar added to A's stored adjoints.
            dy_x, dn_x = dy[E:].clone(), (dn[E:].clone() if use_geo else None)
            ops.step_backward_inputs(2, *common, d_diff, None, st.view_sorted, st.n_eval, st.true_rows, d_eo, d_gth, d_si, st.d_mask, st.e_mask, dy, dn)
            pair = ops.sdf_backward_pair(net, M, M, Mb, dy, dn, E, N, dy_x, dn_x, st.saved)
            if pair is not None:
                wsA, dx = pair
                fbar = ops.step_backward_fbar(st.n_eik, st.n_ds, N, Nout, din, use_geo, d_diff, dx, st.view_sorted, st.n_eval, dy)
                res = ops.sdf_backward_finish(net, M, M, Mb, dy, st.saved, wsA, E, N, fbar, out=out_s)
                dWs, dbs = res if res is not None else ((), ())
            else:                                                # network too wide for the fused chains: the sequential route
                _, _, dx = ops.sdf_backward(net, st.x_eval, M, M, N, dy_x, dn_x, st.saved, True, want_dw=False, row0=E)
                ops.step_backward_fbar(st.n_eik, st.n_ds, N, Nout, din, use_geo, d_diff, dx, st.view_sorted, st.n_eval, dy)
        else:
            ops.step_backward_inputs(1, *common, d_diff, None, st.view_sorted, st.n_eval, st.true_rows, d_eo, d_gth, d_si, st.d_mask, st.e_mask,
                                     dy, dn)
        if dWs is None:
            dWs, dbs, _ = ops.sdf_backward(net, st.x_eval, M, M, Mb, dy, dn, st.saved, False, out=out_s)
        if plan is not None:
            return None, dflat
        if dWr is None:
            dWr = [torch.zeros_like(L.w) for L in rnet.layers]
            dbr = [torch.zeros_like(L.bias) for L in rnet.layers]
        return (None,) + tuple(dWs) + tuple(dbs) + tuple(dWr) + tuple(dbr)


def idr_step(state, ws, bs, rws, rbs):
    return _IdrStep.apply(state, *ws, *bs, *rws, *rbs)


def idr_step_flat(state, flat):
    """The same node with the folded parameters of both networks as one tensor (fold_networks_flat); state.plan must be set."""
    return _IdrStep.apply(state, flat)
