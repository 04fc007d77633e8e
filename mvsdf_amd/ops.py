"""Thin torch-tensor wrappers over the C ABI (raw pointers + current HIP stream).  Device memory comes from
torch; nothing here computes -- every function is one or a few C calls into libmvsdf_hip.so."""
import ctypes as C

import torch

from ._lib import NetDesc, TraceParams, check, lib, ptr, stream_of


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda, 'expected a float32 CUDA(HIP) tensor'
    return t.contiguous()


class PackedLayer:
    __slots__ = ('w', 'wp', 'wpT', 'bias', 'K', 'N', 'wp16', 'wp_ptr', 'wpT_ptr', 'w_ptr', 'keep', 'wx3', 'wx3T')


class PackedNet:
    """A weight-norm-folded MLP: row-major W (for autograd bookkeeping), MFMA-packed W and W^T, biases."""

    def __init__(self, layers, skip_layer, multires):
        """skip_layer: the layer whose input is cat([x, PE]) / sqrt(2) (-1: none), or a sequence of such layers (skip_in, idr.py:46,86)."""
        self.skip_layers = tuple(sorted(int(s) for s in skip_layer)) if isinstance(skip_layer, (tuple, list)) else ((int(skip_layer),) if skip_layer >= 0 else ())
        self.layers, self.multires = layers, multires
        self.skip_layer = self.skip_layers[0] if self.skip_layers else -1
        self.trace_dtype = 0                # 2: the tracing MLP runs on fp32 packs of the bf16-rounded weights; 3 / 4: bf16 packs, activations as 2 / 3 bf16 terms; 5: three-term packs of the fp32 weights

    def desc(self, transposed=False):
        """ctypes descriptor (cached: a PackedNet is immutable once its packs exist; pack_bf16_net drops the cache)."""
        key = ('_dT' if transposed else '_d')
        c = self.__dict__.get(key)
        if c is not None:
            return c
        d = NetDesc()
        d.n_layers = len(self.layers)
        for i, L in enumerate(self.layers):
            d.K[i], d.N[i] = (L.N, L.K) if transposed else (L.K, L.N)
            pk = L.wpT if transposed else L.wp
            d.wp[i] = pk.data_ptr() if pk is not None else (L.wpT_ptr if transposed else L.wp_ptr)      # raw pointers: packs living in one flat buffer
            d.bias[i] = L.bias.data_ptr()
            d.w[i] = L.w.data_ptr() if L.w is not None else L.w_ptr
        d.skip_layer, d.multires = self.skip_layer, self.multires
        if len(self.skip_layers) > 1 and not transposed:
            d.skip_mask = sum(1 << s for s in self.skip_layers)
        for i, L in enumerate(self.layers):                     # three-term packs of W_l / W_l^T: the differentiable chains on the bf16 matrix cores (csrc/chain_x3.h)
            x3 = getattr(L, 'wx3T' if transposed else 'wx3', None)
            if x3 is not None:
                d.wx3[i] = x3 if isinstance(x3, int) else x3.data_ptr()
        if not transposed and self.trace_dtype in (1, 2, 3, 4, 5):
            for i, L in enumerate(self.layers):
                d.wp16[i] = L.wp16.data_ptr()
            d.trace_dtype = self.trace_dtype
        self.__dict__[key] = d
        return d

    def wsizes(self):
        return [L.N * L.K for L in self.layers], [L.N for L in self.layers]


def fold_pack(v, g, want_t=True):
    """weight_norm fold + packing: v[N,K], g[N,1] -> (w[N,K], wp, wpT)  (idr.py:70-71)."""
    v, g = _f32(v), _f32(g).reshape(-1)
    N, K = v.shape
    w = torch.empty_like(v)
    wp = torch.empty(lib().mvsdf_packed_floats(N, K), dtype=torch.float32, device=v.device)
    # the transposed pack pads the OTHER dimension to 32: its size is packed_floats(K, N), not (N, K)
    wpT = torch.empty(lib().mvsdf_packed_floats(K, N), dtype=torch.float32, device=v.device) if want_t else None
    check(lib().mvsdf_fold_pack(ptr(v), ptr(g), N, K, ptr(w), ptr(wp), ptr(wpT), stream_of(v)), 'mvsdf_fold_pack')
    return w, wp, wpT


def fold_backward(v, g, dW):
    v, g, dW = _f32(v), _f32(g).reshape(-1), _f32(dW)
    N, K = v.shape
    dv, dg = torch.empty_like(v), torch.empty_like(g)
    check(lib().mvsdf_fold_backward(ptr(v), ptr(g), ptr(dW), N, K, ptr(dv), ptr(dg), stream_of(v)), 'mvsdf_fold_backward')
    return dv, dg.reshape(-1, 1)


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t_ in enumerate(tensors):
        arr[i] = t_.data_ptr() if t_ is not None else None
    return arr


def fold_pack_net(vs, gs):
    """All layers in one C call -> lists (w, wp, wpT).  gs[l] None: no weight norm for that layer (w = v)."""
    vs = [_f32(v) for v in vs]
    gs = [_f32(g).reshape(-1) if g is not None else None for g in gs]
    dev = vs[0].device
    n = len(vs)
    N = (C.c_int * n)(*[v.shape[0] for v in vs])
    K = (C.c_int * n)(*[v.shape[1] for v in vs])
    ws = [torch.empty_like(v) for v in vs]
    wps = [torch.empty(lib().mvsdf_packed_floats(v.shape[0], v.shape[1]), dtype=torch.float32, device=dev) for v in vs]
    wpTs = [torch.empty(lib().mvsdf_packed_floats(v.shape[1], v.shape[0]), dtype=torch.float32, device=dev) for v in vs]
    check(lib().mvsdf_fold_pack_net(n, _ptr_array(vs), _ptr_array(gs), N, K, _ptr_array(ws), _ptr_array(wps), _ptr_array(wpTs),
                                    stream_of(vs[0])), 'mvsdf_fold_pack_net')
    return ws, wps, wpTs


def _int_ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))()
    for i, v in enumerate(ptrs):
        arr[i] = v
    return arr


class FoldPlan:
    """Everything about folding a fixed set of layers that does not change from step to step: dims, offsets of each layer inside ONE flat
    buffer [W of net 0 | b of net 0 | W of net 1 | b of net 1 ...] (the order in which the backward kernels emit dW_cat / db_cat per network),
    offsets of the MFMA packs inside one flat pack buffer, ctypes arrays.  Built once per model (functional.fold_networks_flat)."""

    def __init__(self, shapes, cuts):
        self.n = len(shapes)
        self.N = (C.c_int * self.n)(*[s[0] for s in shapes])
        self.K = (C.c_int * self.n)(*[s[1] for s in shapes])
        self.shapes = shapes
        self.woff, self.boff, self.seg = [0] * self.n, [0] * self.n, []
        off, lo = 0, 0
        for hi in cuts:                                          # per network: weights, then biases
            w0 = off
            for l in range(lo, hi):
                self.woff[l] = off; off += shapes[l][0] * shapes[l][1]
            b0 = off
            for l in range(lo, hi):
                self.boff[l] = off; off += shapes[l][0]
            self.seg.append((w0, b0, off))
            lo = hi
        self.total = off
        self.poff, self.pToff = [], []
        po = 0
        for (N, K) in shapes:
            self.poff.append(po); po += lib().mvsdf_packed_floats(N, K)
            self.pToff.append(po); po += lib().mvsdf_packed_floats(K, N)
        self.ptotal = po
        self._param_key, self._param_arrays = None, None

    def param_arrays(self, vs, gs, bs):
        """ctypes pointer arrays of the parameters and of their .grad buffers, cached while EVERY storage (parameter and gradient) stays put:
        the key covers all of them, so a gradient detached from the flat buffer in the middle of the list rebuilds the arrays."""
        key = tuple(p.data_ptr() for p in vs) + tuple(p.data_ptr() for p in bs) + tuple(0 if p is None else p.data_ptr() for p in gs) \
            + tuple(0 if p.grad is None else p.grad.data_ptr() for p in vs) + tuple(0 if p.grad is None else p.grad.data_ptr() for p in bs) \
            + tuple(0 if (p is None or p.grad is None) else p.grad.data_ptr() for p in gs)
        if self._param_key != key:
            g_ok = all(p.grad is not None for p in list(vs) + list(bs)) and all(g is None or g.grad is not None for g in gs)
            self._param_arrays = (_ptr_array([v.detach() for v in vs]), _ptr_array([g.detach() if g is not None else None for g in gs]),
                                  (_ptr_array([v.grad for v in vs]), _ptr_array([g.grad if g is not None else None for g in gs]),
                                   _ptr_array([b.grad for b in bs])) if g_ok else None)
            self._param_key = key
        return self._param_arrays


def fold_pack_net_flat(plan, vs, gs, bs, layers):
    """fold + pack of all layers into ONE flat buffer of folded weights (+ room for the bias segment) and ONE buffer of packs: two
    allocations instead of 3 per layer.  Sets L.w_ptr / L.wp_ptr / L.wpT_ptr of `layers`; -> (flat [plan.total], packs)."""
    dev = vs[0].device
    flat = torch.empty(plan.total, dtype=torch.float32, device=dev)
    packs = torch.empty(plan.ptotal, dtype=torch.float32, device=dev)
    fb, pb = flat.data_ptr(), packs.data_ptr()
    wps, wpTs, wss = [], [], []
    for l, L in enumerate(layers):
        L.w = L.wp = L.wpT = None
        L.w_ptr, L.wp_ptr, L.wpT_ptr = fb + 4 * plan.woff[l], pb + 4 * plan.poff[l], pb + 4 * plan.pToff[l]
        wss.append(L.w_ptr); wps.append(L.wp_ptr); wpTs.append(L.wpT_ptr)
    pv, pg, _ = plan.param_arrays(vs, gs, bs)                      # the Parameters themselves (same key as the backward: one cache entry per step)
    check(lib().mvsdf_fold_pack_net(plan.n, pv, pg, plan.N, plan.K, _int_ptr_array(wss), _int_ptr_array(wps), _int_ptr_array(wpTs),
                                    stream_of(vs[0])), 'mvsdf_fold_pack_net')
    return flat, packs


def fold_backward_net_flat(plan, vs, gs, bs, dflat, sink):
    """Backward of fold_pack_net_flat from the flat gradient [dW | db per network].  sink: add dv / dg / db into the parameters' .grad buffers
    (-> None); otherwise -> (dvs, dgs, dbs)."""
    base = dflat.data_ptr()
    dW = _int_ptr_array([base + 4 * o for o in plan.woff])
    db = _int_ptr_array([base + 4 * o for o in plan.boff])
    pv, pg, grads = plan.param_arrays(vs, gs, bs)
    if sink and grads is not None:
        check(lib().mvsdf_fold_backward_net(plan.n, pv, pg, dW, db, plan.N, plan.K, grads[0], grads[1], grads[2], 1, stream_of(dflat)),
              'mvsdf_fold_backward_net')
        return None
    dvs = [torch.empty_like(v) for v in vs]
    dgs = [torch.empty_like(g) if g is not None else None for g in gs]
    check(lib().mvsdf_fold_backward_net(plan.n, pv, pg, dW, None, plan.N, plan.K, _ptr_array(dvs), _ptr_array(dgs), None, 0, stream_of(dflat)),
          'mvsdf_fold_backward_net')
    dbs = [dflat[o:o + s[0]] for o, s in zip(plan.boff, plan.shapes)]
    return dvs, dgs, dbs


def fold_backward_net(vs, gs, dWs, dbs=None, sinks=None):
    """-> (dvs, dgs).  With `sinks` = (dv_targets, dg_targets, db_targets) the results (and the bias gradients dbs) are ADDED into
    those tensors instead (the parameters' .grad buffers) and nothing is returned.  gs[l] None (no weight norm): dv = dW, dg None."""
    vs = [_f32(v) for v in vs]
    gs = [_f32(g).reshape(-1) if g is not None else None for g in gs]
    dWs = [_f32(d) for d in dWs]
    n = len(vs)
    N = (C.c_int * n)(*[v.shape[0] for v in vs])
    K = (C.c_int * n)(*[v.shape[1] for v in vs])
    if sinks is not None:
        tv, tg, tb = sinks
        for t in list(tv) + list(tg) + list(tb):
            assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous())
        dbs = [_f32(d) for d in dbs]
        check(lib().mvsdf_fold_backward_net(n, _ptr_array(vs), _ptr_array(gs), _ptr_array(dWs), _ptr_array(dbs), N, K, _ptr_array(tv),
                                            _ptr_array(tg), _ptr_array(tb), 1, stream_of(vs[0])), 'mvsdf_fold_backward_net')
        return None
    dvs = [torch.empty_like(v) for v in vs]
    dgs = [torch.empty_like(g) if g is not None else None for g in gs]
    check(lib().mvsdf_fold_backward_net(n, _ptr_array(vs), _ptr_array(gs), _ptr_array(dWs), None, N, K, _ptr_array(dvs), _ptr_array(dgs),
                                        None, 0, stream_of(vs[0])), 'mvsdf_fold_backward_net')
    return dvs, [g.reshape(-1, 1) if g is not None else None for g in dgs]


CHAIN_X3 = True          # SDF networks get the three-term packs of W_l and W_l^T: their differentiable chains run on the bf16 matrix cores (csrc/chain_x3.h)


def pack_x3_chain(net):
    """Three-term bf16 packs of every W_l and W_l^T of a folded SDF network (MvsdfNetDesc.wx3 of both descriptors)."""
    n = len(net.layers)
    dev = net.layers[0].bias.device
    N = (C.c_int * n)(*[L.N for L in net.layers])
    K = (C.c_int * n)(*[L.K for L in net.layers])
    ws = _int_ptr_array([L.w.data_ptr() if L.w is not None else L.w_ptr for L in net.layers])
    for L in net.layers:
        L.wx3 = torch.empty(3 * lib().mvsdf_packed_bf16_bytes(L.N, L.K, 0), dtype=torch.uint8, device=dev)
        L.wx3T = torch.empty(3 * lib().mvsdf_packed_bf16_bytes(L.K, L.N, 0), dtype=torch.uint8, device=dev)
    s = stream_of(net.layers[0].bias)
    check(lib().mvsdf_pack_bf16x3_net(n, ws, N, K, _ptr_array([L.wx3 for L in net.layers]), s), 'mvsdf_pack_bf16x3_net')
    check(lib().mvsdf_pack_bf16x3t_net(n, ws, N, K, _ptr_array([L.wx3T for L in net.layers]), s), 'mvsdf_pack_bf16x3t_net')
    net.__dict__.pop('_d', None)
    net.__dict__.pop('_dT', None)
    return net


def pack_net(vs, gs, biases, skip_layer, multires, want_t=True, x3=None):
    layers = []
    for v, g, b in zip(vs, gs, biases):
        L = PackedLayer()
        L.w, L.wp, L.wpT = fold_pack(v, g, want_t)
        L.bias = _f32(b)
        L.N, L.K = v.shape
        L.wp16 = None
        L.wx3 = L.wx3T = None
        layers.append(L)
    net = PackedNet(layers, skip_layer, multires)
    if (CHAIN_X3 if x3 is None else x3) and want_t:
        maybe_pack_x3_chain(net, force=x3)
    return net


def maybe_pack_x3_chain(net, force=None):
    """pack_x3_chain for SDF networks (first Linear over the positional encoding of a 3-D point); the rendering network's chains stay on the fp32-input MFMA.
    force: pack_net's `x3` argument (None: the module default CHAIN_X3 decides)."""
    if (CHAIN_X3 if force is None else force) and len(net.layers) >= 2 and net.layers[0].K == 3 + 6 * max(net.multires, 0):
        pack_x3_chain(net)
    return net


TRACE_DTYPES = {'f32': 0, 'bf16w': 2, 'bf16x2': 3, 'bf16x3': 4, 'f32x3': 5}      # (1, 'bf16' -- 8-bit activations too -- was removed in round 5: 'bf16x2' dominates it)


def pack_trace_net(net, dtype):
    """Switch the tracing MLP of a folded SDF network to one of TRACE_DTYPES (IDRNetwork.set_trace_dtype)."""
    if dtype == 'f32':
        return net
    return pack_bf16_net(net, weights_only=(dtype == 'bf16w'), terms={'bf16x2': 2, 'bf16x3': 3, 'f32x3': 3}.get(dtype, 0), weight_terms=3 if dtype == 'f32x3' else 1)


def pack_bf16_net(net, weights_only=False, terms=0, weight_terms=1):
    """bf16 MFMA packs of a folded SDF network (BASELINE configs[4], csrc/tile_engine_bf16s.h): one launch; switches the network's tracing
    MLP (ops.trace, ops.sdf_col0) to one of the bf16-matrix-core arithmetics.  The differentiable passes keep the fp32 weights.
    weights_only: only the WEIGHTS are rounded to bf16 (fp32 packs of the rounded values, fp32 activations on the fp32 MFMA, trace_dtype 2):
    bit-exact against the oracle on the rounded weights.
    terms = 2 / 3: bf16 weights on the bf16 MFMA, every activation carried as 2 / 3 bf16 terms (16 / all 24 mantissa bits;
    csrc/tile_engine_bf16s.h, trace_dtype 3 / 4): the arithmetic of weights_only up to the order of the fp32 additions, at bf16-MFMA speed.
    terms = 3, weight_terms = 3 ('f32x3', trace_dtype 5): the fp32 weights UNROUNDED, as three bf16 terms like the activations -- the reference's fp32
    Linear (idr.py:89) from six exact bf16 products per element pair on the bf16 MFMA; fp32-accurate (closer to an fp64 evaluation than the fp32
    fmaf chain) and bit-exact against oracle.Net(sd, bf16='f32x3') (a model of the matrix instruction); not bit-identical to 'f32'."""
    n = len(net.layers)
    if terms:
        assert terms in (2, 3) and not weights_only and weight_terms in (1, 3) and (weight_terms == 1 or terms == 3)
        dev = net.layers[0].bias.device
        for L in net.layers:
            L.wp16 = torch.empty(weight_terms * lib().mvsdf_packed_bf16_bytes(L.N, L.K, 0), dtype=torch.uint8, device=dev)
        N = (C.c_int * n)(*[L.N for L in net.layers])
        K = (C.c_int * n)(*[L.K for L in net.layers])
        fn = lib().mvsdf_pack_bf16x3_net if weight_terms == 3 else lib().mvsdf_pack_bf16s_net
        check(fn(n, _int_ptr_array([L.w.data_ptr() if L.w is not None else L.w_ptr for L in net.layers]), N, K,
                 _ptr_array([L.wp16 for L in net.layers]), stream_of(net.layers[0].bias)), 'mvsdf_pack_bf16s_net / mvsdf_pack_bf16x3_net')
        net.trace_dtype = 5 if weight_terms == 3 else 1 + terms
        net.__dict__.pop('_d', None)
        return net
    if weights_only:
        dev = net.layers[0].bias.device
        for L in net.layers:
            L.wp16 = torch.empty(lib().mvsdf_packed_floats(L.N, L.K), dtype=torch.float32, device=dev)
        N = (C.c_int * n)(*[L.N for L in net.layers])
        K = (C.c_int * n)(*[L.K for L in net.layers])
        check(lib().mvsdf_pack_bf16w_net(n, _int_ptr_array([L.w.data_ptr() if L.w is not None else L.w_ptr for L in net.layers]), N, K,
                                         _ptr_array([L.wp16 for L in net.layers]), stream_of(net.layers[0].bias)), 'mvsdf_pack_bf16w_net')
        net.trace_dtype = 2
        net.__dict__.pop('_d', None)
        return net
    raise ValueError("pack_bf16_net: give weights_only=True ('bf16w') or terms=2 / 3 ('bf16x2' / 'bf16x3'; with weight_terms=3: 'f32x3') -- the engine that rounded the "
                     "hidden activations to bf16 too (trace_dtype 1) was removed in round 5")


def sdf_col0(net, x, mt=2):
    x = _f32(x)
    y = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    d = net.desc()
    check(lib().mvsdf_sdf_col0(C.byref(d), ptr(x), x.shape[0], ptr(y), mt, stream_of(x)), 'mvsdf_sdf_col0')
    return y


def camera_rays(uv, pose, intrinsics):
    uv, pose, intrinsics = _f32(uv), _f32(pose), _f32(intrinsics)
    B, P = uv.shape[:2]
    dirs = torch.empty(B, P, 3, dtype=torch.float32, device=uv.device)
    cam = torch.empty(B, 3, dtype=torch.float32, device=uv.device)
    check(lib().mvsdf_camera_rays(ptr(uv), ptr(pose), ptr(intrinsics), B, P, ptr(dirs), ptr(cam), stream_of(uv)), 'mvsdf_camera_rays')
    return dirs, cam


def sphere_intersection(cam_loc, ray_dirs, r=1.0):
    cam_loc, ray_dirs = _f32(cam_loc), _f32(ray_dirs)
    B, P = ray_dirs.shape[:2]
    t = torch.empty(B, P, 2, dtype=torch.float32, device=ray_dirs.device)
    m = torch.empty(B, P, dtype=torch.uint8, device=ray_dirs.device)
    check(lib().mvsdf_sphere_intersection(ptr(cam_loc), ptr(ray_dirs), B, P, C.c_float(r), ptr(t), ptr(m), stream_of(ray_dirs)),
          'mvsdf_sphere_intersection')
    return t, m.bool()


_MINSDF_SIDE_STREAM = False                         # trace stages 5 / 6: the min-sdf rows on a second stream under the sampler launches (measured slower; tests set it)
_side_streams = {}


def _minsdf_stream(dev):
    s = _side_streams.get(dev)
    if s is None:
        s = _side_streams[dev] = torch.cuda.Stream(dev)
    return s


def trace(net, cam_loc, ray_dirs, object_mask, params, training, intervals, minsdf_steps=None, mt=1, mt_samples=4, events=None,
          mask_ready=None, defer_minsdf=None):
    """RayTracing.forward on the device -> (points[R,3], mask[R] bool, dists[R], counters[16] int64 device tensor).
    mt: row tiles per sphere-tracing workgroup (8*mt rays); mt_samples: row tiles per chunk of the sample-row kernels.
    events: optional list; when given the two kernels are launched by separate C calls and (start, mid, end) torch events
    recorded on the current stream are appended (per-kernel timing for bench.py's roofline).
    mask_ready: optional callable(mask_bool) invoked (on the host) after the launch that finalises the hit mask and BEFORE the secant /
    min-sdf launch is enqueued: whatever it enqueues (e.g. an async copy of the hit count) completes while that last launch runs.
    defer_minsdf: optional list; when given (with mask_ready, training) the min-sdf rows are NOT evaluated: a callable that evaluates them
    later (stage 5: it rewrites points / dists of the non-hit rays in place) is appended instead."""
    cam_loc, ray_dirs = _f32(cam_loc), _f32(ray_dirs)
    B, P = ray_dirs.shape[:2]
    R = B * P
    dev = ray_dirs.device
    om = object_mask.reshape(-1).contiguous()
    om = om.view(torch.uint8) if om.dtype == torch.bool else om.to(torch.uint8)       # bool storage is one byte 0/1: no copy
    pts = torch.empty(R, 3, dtype=torch.float32, device=dev)
    mask = torch.empty(R, dtype=torch.uint8, device=dev)
    dists = torch.empty(R, dtype=torch.float32, device=dev)
    counters = torch.empty(16, dtype=torch.int64, device=dev)
    tp = TraceParams(*params)
    wsb = lib().mvsdf_trace_workspace_bytes_n(R, tp.n_steps)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    d = net.desc()
    iv = _f32(intervals)
    st = _f32(minsdf_steps) if minsdf_steps is not None else None
    args = (C.byref(d), C.byref(tp), ptr(cam_loc), ptr(ray_dirs), ptr(om), B, P, 1 if training else 0, ptr(iv), ptr(st),
            ptr(pts), ptr(mask), ptr(dists), ptr(counters), ptr(ws), C.c_size_t(wsb), mt, mt_samples, stream_of(ray_dirs))
    if mask_ready is not None:
        # events: (start, after sphere tracing, after the sampler rows, before secant / min-sdf, end)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if events is not None else None
        if ev: ev[0].record()
        check(lib().mvsdf_trace_stage(1, *args), 'mvsdf_trace_stage(1)')
        if ev: ev[1].record()
        side = _minsdf_stream(dev) if (training and _MINSDF_SIDE_STREAM) else None
        if side is not None:
            # the min-sdf rows need only the sphere tracer's work list: they run on a second stream under the (latency-shaped) sampler launches
            main = torch.cuda.current_stream(dev)
            e_in, e_out = torch.cuda.Event(), torch.cuda.Event()
            e_in.record(main)
            side.wait_event(e_in)
            sargs = args[:-1] + (C.c_void_p(side.cuda_stream),)
            check(lib().mvsdf_trace_stage(5, *sargs), 'mvsdf_trace_stage(5)')
            e_out.record(side)
        check(lib().mvsdf_trace_stage(3, *args), 'mvsdf_trace_stage(3)')
        if ev: ev[2].record()
        mask_b = mask.view(torch.bool)                           # the kernels write 0 / 1 bytes
        mask_ready(mask_b)
        if ev: ev[3].record()
        if side is not None:
            check(lib().mvsdf_trace_stage(6, *args), 'mvsdf_trace_stage(6)')
            torch.cuda.current_stream(dev).wait_event(e_out)
        elif defer_minsdf is not None and training:
            check(lib().mvsdf_trace_stage(6, *args), 'mvsdf_trace_stage(6)')
            keep = (net, cam_loc, ray_dirs, om, iv, st, pts, mask, dists, counters, ws, d, tp)      # everything `args` points into

            def finish(keep=keep, args=args):
                a = args[:-1] + (stream_of(keep[2]),)
                check(lib().mvsdf_trace_stage(5, *a), 'mvsdf_trace_stage(5)')
            defer_minsdf.append(finish)
        else:
            check(lib().mvsdf_trace_stage(4, *args), 'mvsdf_trace_stage(4)')
        if ev:
            ev[4].record()
            events.append(tuple(ev))
        return pts, mask_b, dists, counters
    if events is None:
        check(lib().mvsdf_trace(*args), 'mvsdf_trace')
    else:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        check(lib().mvsdf_trace_stage(1, *args), 'mvsdf_trace_stage(1)')
        ev[1].record()
        check(lib().mvsdf_trace_stage(2, *args), 'mvsdf_trace_stage(2)')
        ev[2].record()
        events.append(tuple(ev))
    return pts, mask.view(torch.bool), dists, counters


def det_math(op, x):
    x = _f32(x).reshape(-1)
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    check(lib().mvsdf_det_math(op, ptr(x), x.numel(), ptr(y0), ptr(y1), stream_of(x)), 'mvsdf_det_math')
    return y0, y1


def _split_cat(net, dW_cat, db_cat):
    ws, bs = net.wsizes()
    dWs = [t.view(L.N, L.K) for t, L in zip(torch.split(dW_cat, ws), net.layers)]
    dbs = list(torch.split(db_cat, bs))
    return dWs, dbs


def sdf_forward(net, x, Mg):
    """value + normal: x[M,3] -> y[M,Nout], n[Mg,3] (first Mg rows), ctx (saved activations)."""
    x = _f32(x)
    M, dev = x.shape[0], x.device
    d, dT = net.desc(), net.desc(True)
    y = torch.empty(M, net.layers[-1].N, dtype=torch.float32, device=dev)
    n = torch.empty(Mg, 3, dtype=torch.float32, device=dev)
    ctx = torch.empty(lib().mvsdf_sdf_ctx_floats(C.byref(d), M, Mg), dtype=torch.float32, device=dev)
    check(lib().mvsdf_sdf_forward(C.byref(d), C.byref(dT), ptr(x), M, Mg, ptr(y), ptr(n) if Mg > 0 else None, ptr(ctx), stream_of(x)),
          'mvsdf_sdf_forward')
    return y, n, ctx


def _check_rows(name, net, Mb, dy, dn):
    """The kernels index dy / dn by row with the network's own output width: a tensor of another shape would be read as garbage, not rejected."""
    nout = net.layers[-1].N
    if dy.dim() != 2 or dy.shape[0] < Mb or dy.shape[1] != nout:
        raise ValueError('%s: dy must be [>= %d, %d], got %s' % (name, Mb, nout, tuple(dy.shape)))
    if dn is not None and (dn.dim() != 2 or dn.shape[0] < Mb or dn.shape[1] != 3):
        raise ValueError('%s: dn must be [>= %d, 3], got %s' % (name, Mb, tuple(dn.shape)))


def sdf_backward(net, x, M, Mg, Mb, dy, dn, ctx, want_dx, want_dw=True, row0=0, out=None):
    """-> (dWs [list per layer], dbs, dx or None) over rows [row0, row0 + Mb) (dWs = dbs = None when want_dw is False).
    x is the full [M,3] point tensor of the forward; dy / dn hold Mb rows."""
    x, dy = _f32(x), _f32(dy)
    dev = x.device
    _check_rows('sdf_backward', net, Mb, dy, dn)
    d, dT = net.desc(), net.desc(True)
    ws_n, bs_n = net.wsizes()
    if out is not None:                                          # (dW_cat, db_cat) slices of a flat gradient: written in place, not split
        dW, db = out
    else:
        dW = torch.empty(sum(ws_n), dtype=torch.float32, device=dev) if want_dw else None
        db = torch.empty(sum(bs_n), dtype=torch.float32, device=dev) if want_dw else None
    dx = torch.empty(Mb, 3, dtype=torch.float32, device=dev) if want_dx else None
    ws = torch.empty(lib().mvsdf_sdf_bwd_ws_floats(C.byref(d), Mb), dtype=torch.float32, device=dev)
    dn = _f32(dn) if dn is not None else None
    xr = x[row0:] if row0 else x
    check(lib().mvsdf_sdf_backward(C.byref(d), C.byref(dT), ptr(xr), M, Mg, row0, Mb, ptr(dy), ptr(dn), ptr(ctx), ptr(dW), ptr(db), ptr(dx),
                                   ptr(ws), stream_of(x)), 'mvsdf_sdf_backward')
    if not want_dw or out is not None:
        return None, None, dx
    dWs, dbs = _split_cat(net, dW, db)
    return dWs, dbs, dx


def sdf_backward_pair(net, M, Mg, MbA, dyA, dnA, row0X, MbX, dyX, dnX, ctx):
    """Pass A (full backward over rows [0, MbA), adjoints kept in the returned workspace) and pass X (input adjoint of rows
    [row0X, row0X + MbX) for the upstream (dyX, dnX)) as ONE grid.  -> (wsA, dx[MbX,3]) or None when the fused chains do not cover the net."""
    dev = dyA.device
    _check_rows('sdf_backward_pair (A)', net, MbA, dyA, dnA)
    _check_rows('sdf_backward_pair (X)', net, MbX, dyX, dnX)
    d, dT = net.desc(), net.desc(True)
    wsA = torch.empty(lib().mvsdf_sdf_bwd_ws_floats(C.byref(d), MbA), dtype=torch.float32, device=dev)
    wsX = torch.empty(lib().mvsdf_sdf_bwd_ws_floats(C.byref(d), MbX), dtype=torch.float32, device=dev)
    dx = torch.empty(MbX, 3, dtype=torch.float32, device=dev)
    rc = lib().mvsdf_sdf_backward_pair(C.byref(d), C.byref(dT), M, Mg, MbA, ptr(_f32(dyA)), ptr(_f32(dnA)), ptr(wsA), row0X, MbX, ptr(_f32(dyX)),
                                       ptr(_f32(dnX)) if dnX is not None else None, ptr(wsX), ptr(dx), ptr(ctx), stream_of(dyA))
    if rc == -3:
        return None
    check(rc, 'mvsdf_sdf_backward_pair')
    return wsA, dx


def sdf_backward_finish(net, M, Mg, Mb, dy, ctx, wsA, row0D, MbD, fbar, out=None):
    """Delta pass (fbar on output column 0 of rows [row0D, row0D + MbD), added to the stored adjoints) + weight gradients -> (dWs, dbs)."""
    dev = dy.device
    _check_rows('sdf_backward_finish', net, Mb, dy, None)
    d, dT = net.desc(), net.desc(True)
    if out is not None:
        dW, db = out
    else:
        ws_n, bs_n = net.wsizes()
        dW = torch.empty(sum(ws_n), dtype=torch.float32, device=dev)
        db = torch.empty(sum(bs_n), dtype=torch.float32, device=dev)
    check(lib().mvsdf_sdf_backward_finish(C.byref(d), C.byref(dT), M, Mg, Mb, ptr(dy), ptr(ctx), ptr(wsA), row0D, MbD, ptr(fbar) if MbD > 0 else None,
                                          ptr(dW), ptr(db), stream_of(dy)), 'mvsdf_sdf_backward_finish')
    return None if out is not None else _split_cat(net, dW, db)


def step_backward_fbar(n_eik, n_ds, N, Nout, din, use_geo, d_diff, dx, view_sorted, n_eval, dy):
    """SampleNetwork's scalar per hit row (sample_network.py:10-20 backward): -> fbar[N]; also added to dy[(E + i), 0]."""
    o = lambda t: None if t is None else ptr(_f32(t))
    fbar = torch.empty(N, dtype=torch.float32, device=dy.device)
    check(lib().mvsdf_step_backward_fbar(n_eik, n_ds, N, Nout, o(din), din.shape[1] if din is not None else 0, 1 if use_geo else 0, o(d_diff), o(dx),
                                         ptr(view_sorted), ptr(n_eval), ptr(dy), ptr(fbar), stream_of(dy)), 'mvsdf_step_backward_fbar')
    return fbar


def render_forward(net, points, view, normals, feat, multires_view):
    """feat may be a column slice of a wider row-major tensor (stride(0) = ld)."""
    points, view, normals = _f32(points), _f32(view), _f32(normals)
    assert feat.dtype == torch.float32 and feat.is_cuda and feat.stride(1) == 1
    N, dev = points.shape[0], points.device
    d = net.desc()
    rgb = torch.empty(N, net.layers[-1].N, dtype=torch.float32, device=dev)
    ctx = torch.empty(lib().mvsdf_render_ctx_floats(C.byref(d), N), dtype=torch.float32, device=dev)
    check(lib().mvsdf_render_forward(C.byref(d), ptr(points), ptr(view), ptr(normals), C.c_void_p(feat.data_ptr()), feat.stride(0), N,
                                     multires_view, ptr(rgb), ptr(ctx), stream_of(points)), 'mvsdf_render_forward')
    return rgb, ctx


def render_backward(net, N, drgb, ctx, n_ctx=None, out=None):
    """n_ctx: rows the forward context was made with (default N); the backward covers its first N rows."""
    drgb = _f32(drgb)
    dev = drgb.device
    d, dT = net.desc(), net.desc(True)
    if out is not None:
        dW, db = out
    else:
        ws_n, bs_n = net.wsizes()
        dW = torch.empty(sum(ws_n), dtype=torch.float32, device=dev)
        db = torch.empty(sum(bs_n), dtype=torch.float32, device=dev)
    din = torch.empty(N, net.layers[0].K, dtype=torch.float32, device=dev)
    ws = torch.empty(lib().mvsdf_render_bwd_ws_floats(C.byref(d), N), dtype=torch.float32, device=dev)
    check(lib().mvsdf_render_backward(C.byref(d), C.byref(dT), N, N if n_ctx is None else n_ctx, ptr(drgb), ptr(ctx), ptr(dW), ptr(db),
                                      ptr(din), ptr(ws), stream_of(drgb)), 'mvsdf_render_backward')
    if out is not None:
        return None, None, din
    dWs, dbs = _split_cat(net, dW, db)
    return dWs, dbs, din


def feat_corr(pts, view_start, feat, feat_src, cam, src_cams, size, center):
    """-> (loss_pp[N], dpts[N,3]).  feat [B,C,H,W], feat_src [B,V,C,H,W] with any strides (NCHW / channels_last)."""
    pts = _f32(pts)
    N, dev = pts.shape[0], pts.device
    B, Cc, H, W = feat.shape
    V = feat_src.shape[1]
    assert feat.dtype == torch.float32 and feat_src.dtype == torch.float32 and feat.is_cuda and feat_src.is_cuda
    fs = (C.c_longlong * 4)(*feat.stride())
    ss = (C.c_longlong * 5)(*feat_src.stride())
    loss_pp = torch.empty(N, dtype=torch.float32, device=dev)
    dpts = torch.empty(N, 3, dtype=torch.float32, device=dev)
    vs = view_start.to(torch.int32).contiguous()
    check(lib().mvsdf_feat_corr(ptr(pts), N, ptr(vs), B, V, Cc, H, W, C.c_void_p(feat.data_ptr()), fs, C.c_void_p(feat_src.data_ptr()), ss,
                                ptr(_f32(cam)), ptr(_f32(src_cams)), ptr(_f32(size).reshape(-1)), ptr(_f32(center).reshape(-1)),
                                ptr(loss_pp), ptr(dpts), stream_of(pts)), 'mvsdf_feat_corr')
    return loss_pp, dpts


def depth_carve(pts, depths, cams, size, center, out_thresh_perc, far_thresh, far_att, near_thresh, near_att, world_inplace=False, use_invalid=False):
    """pts [M,3] or [M,4] (hom, contiguous) normalised; depths [B,h,w]; cams [B,2,4,4] -> (dist_r[M], weight[M]).
    world_inplace: also overwrite pts[:, :3] with the world-space points (the reference's side effect, loss.py:38,42).
    use_invalid: carving_t instead of carving_t2 (conf.use_invalid, loss.py:43-46)."""
    pts, depths, cams = _f32(pts), _f32(depths), _f32(cams)
    M, dev = pts.shape[0], pts.device
    B, h, w = depths.shape
    dist_r = torch.empty(M, dtype=torch.float32, device=dev)
    weight = torch.empty(M, dtype=torch.float32, device=dev)
    check(lib().mvsdf_depth_carve(ptr(pts), pts.shape[1], M, ptr(depths), B, h, w, ptr(cams), ptr(_f32(size).reshape(-1)),
                                  ptr(_f32(center).reshape(-1)), C.c_float(out_thresh_perc), C.c_float(far_thresh), C.c_float(far_att),
                                  C.c_float(near_thresh), C.c_float(near_att), 1 if use_invalid else 0, ptr(dist_r), ptr(weight), ptr(pts) if world_inplace else None,
                                  stream_of(pts)), 'mvsdf_depth_carve')
    return dist_r, weight


def loss_terms(rgb, rgb_gt, rgb_mask, grad_theta, eik_out, dist_r, dweight, surf, n_pos, feat_pp, weights, surf_on, feat_on, inv_counts=None):
    """-> (out[6] = loss, rgb, eikonal, depth, feat, surf; d_rgb, d_grad, d_eik_out, d_surf).  One launch.
    inv_counts: optional device float[3] replacing 1/n of the eikonal / depth / surf means (data-parallel exact mode).
    weights = (w_rgb, w_eik, w_surf, w_feat, w_depth[, smooth]): smooth > 0 makes the depth term SmoothL1(eo / smooth, -dist_r / smooth) * smooth (loss.py:57-58)."""
    rgb, rgb_gt = _f32(rgb), _f32(rgb_gt).reshape(-1, 3)
    dev = rgb.device
    R = rgb.shape[0]
    m = rgb_mask.contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
    gt_ = _f32(grad_theta) if grad_theta is not None and grad_theta.shape[0] > 0 else None
    eo = _f32(eik_out).reshape(-1)
    sf = _f32(surf).reshape(-1) if surf is not None else None
    out = torch.empty(6, dtype=torch.float32, device=dev)
    d_rgb = torch.empty_like(rgb)
    d_grad = torch.empty_like(gt_) if gt_ is not None else None
    d_eo = torch.empty_like(eo)
    d_sf = torch.empty_like(sf) if sf is not None else None
    npos = n_pos.to(torch.int64).reshape(1).contiguous() if n_pos is not None else None
    fp = _f32(feat_pp) if feat_pp is not None else None
    w = [float(x) for x in weights] + [0.0]                       # weights[5] (optional): conf.smooth of the depth term, 0 / None = L1
    check(lib().mvsdf_loss_terms(ptr(rgb), ptr(rgb_gt), ptr(m), R, ptr(gt_), gt_.shape[0] if gt_ is not None else 0, ptr(eo), ptr(_f32(dist_r)),
                                 ptr(_f32(dweight)), eo.numel(), ptr(sf), sf.numel() if sf is not None else 0, ptr(npos), ptr(fp),
                                 fp.numel() if fp is not None else 0, C.c_float(w[0]), C.c_float(w[1]), C.c_float(w[2]), C.c_float(w[3]),
                                 C.c_float(w[4]), C.c_float(w[5]), 1 if surf_on else 0, 1 if feat_on else 0, ptr(_f32(inv_counts)) if inv_counts is not None else None,
                                 ptr(out), ptr(d_rgb), ptr(d_grad), ptr(d_eo), ptr(d_sf), stream_of(rgb)), 'mvsdf_loss_terms')
    return out, d_rgb, d_grad, d_eo, d_sf


# ---- bookkeeping of one training step (csrc/step_kernels.hip)
def partition_rays(net_mask, object_mask, true_mask, ray_dirs):
    """-> (perm, inv, true_rows, counts[2] device int64, view_sorted[R,3]); see mvsdf_partition_rays."""
    R, dev = net_mask.numel(), net_mask.device
    u8 = lambda m: None if m is None else (m if m.dtype == torch.uint8 else m.view(torch.uint8)).contiguous()
    nm, om, tm = u8(net_mask.reshape(-1)), u8(object_mask), u8(true_mask)
    perm = torch.empty(R, dtype=torch.int64, device=dev)
    inv, true_rows = torch.empty_like(perm), torch.empty_like(perm)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    view = torch.empty(R, 3, dtype=torch.float32, device=dev)
    rd = _f32(ray_dirs.reshape(-1, 3))
    check(lib().mvsdf_partition_rays(ptr(nm), ptr(om), ptr(tm), ptr(rd), R, ptr(perm), ptr(inv), ptr(true_rows), ptr(counts), ptr(view),
                                     stream_of(rd)), 'mvsdf_partition_rays')
    return perm, inv, true_rows, counts, view


def step_outputs(R, n_eik, n_ds, counts, x_eval, y_eval, n_eval, inv, true_rows, rgb_sorted, d_mask, e_mask):
    """Worst-case sized outputs (N = R); the caller narrows them once the host knows N / n_true (see mvsdf_step_outputs).
    -> rgb_values [R,3], sdf_output [R,1], diff_pts [R,3], eik_out [R+E], points_hom [R+E,4], grad_theta [R+E,3], surf [R+n_eik]"""
    dev, Nout, E = x_eval.device, y_eval.shape[1], n_eik + 2 * n_ds
    f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    rgb_values, sdf_output, diff_pts = f(R, 3), f(R, 1), f(R, 3)
    eik_out, hom, gth, surf = f(R + E), f(R + E, 4), f(R + E, 3), f(R + n_eik)
    check(lib().mvsdf_step_outputs(R, n_eik, n_ds, Nout, ptr(counts), ptr(x_eval), ptr(y_eval), ptr(n_eval), ptr(inv), ptr(true_rows),
                                   ptr(rgb_sorted), d_mask, e_mask, ptr(rgb_values), ptr(sdf_output), ptr(diff_pts), ptr(eik_out), ptr(hom),
                                   ptr(gth), ptr(surf), stream_of(x_eval)), 'mvsdf_step_outputs')
    return rgb_values, sdf_output, diff_pts, eik_out, hom, gth, surf


def step_backward_inputs(stage, n_eik, n_ds, N, Nout, n_true, din, din_feat0, din_nrm0, use_geo, d_diff, dx, view_sorted, n_eval, true_rows,
                         d_eo, d_gth, d_si, d_mask, e_mask, dy, dn):
    o = lambda t: None if t is None else ptr(_f32(t))
    check(lib().mvsdf_step_backward_inputs(stage, n_eik, n_ds, N, Nout, n_true, o(din), din.shape[1] if din is not None else 0, din_feat0,
                                           din_nrm0, 1 if use_geo else 0, o(d_diff), o(dx), ptr(view_sorted), ptr(n_eval), ptr(true_rows),
                                           o(d_eo), o(d_gth), o(d_si), d_mask, e_mask, ptr(dy), ptr(dn), stream_of(dy)),
          'mvsdf_step_backward_inputs')


# ---- phase-0 depth-surface sampling (csrc/sample_kernels.hip)
def dsurf_samples(depths, depth_cams, size, center, bb, jitter_rad, seed, n):
    """depths [N,1,H,W] (or [N,H,W]), depth_cams [N,2,4,4] -> (pts_on [n,3], pts_jit [n,3], counts [2] device int64, idx_sorted [2,n]).
    counts[s] < n means the depth maps hold fewer than n valid in-box pixels for set s (the reference's np.random.choice raises there)."""
    depths = _f32(depths.reshape(depths.shape[0], depths.shape[-2], depths.shape[-1]))
    N, H, W = depths.shape
    dev = depths.device
    kinv = torch.linalg.inv(depth_cams[:, 1, :3, :3]).contiguous()                # my_utils.py:84
    einv = torch.linalg.inv(depth_cams[:, 0]).contiguous()                        # my_utils.py:93
    size, center = _f32(size.reshape(-1)[:1]), _f32(center.reshape(-1)[:3])
    idx = torch.full((2, n), 1 << 62, dtype=torch.int64, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    geo = (ptr(depths), ptr(kinv), ptr(einv), N, H, W, ptr(size), ptr(center), C.c_float(bb), C.c_float(jitter_rad), C.c_ulonglong(seed), n)
    check(lib().mvsdf_dsurf_select(*geo, ptr(idx), ptr(counts), stream_of(depths)), 'mvsdf_dsurf_select')
    idx_sorted = torch.sort(idx, dim=1).values                                    # reference: np.sort(sample_idx)
    pts_on = torch.empty(n, 3, dtype=torch.float32, device=dev)
    pts_jit = torch.empty(n, 3, dtype=torch.float32, device=dev)
    check(lib().mvsdf_dsurf_points(*geo, ptr(idx_sorted), ptr(counts), ptr(pts_on), ptr(pts_jit), stream_of(depths)), 'mvsdf_dsurf_points')
    return pts_on, pts_jit, counts, idx_sorted


# ---- loss bookkeeping (csrc/loss_kernels.hip)
def loss_prep(net_mask, obj_mask, true_mask, B):
    """-> (hit bool [R], view_start int32 [B+1], n_pos int64 scalar tensor); see mvsdf_loss_prep."""
    u8 = lambda m: (m.reshape(-1).contiguous().view(torch.uint8) if m.dtype == torch.bool else m.reshape(-1).to(torch.uint8).contiguous())
    nm, om, tm = u8(net_mask), u8(obj_mask), u8(true_mask)
    R, dev = nm.numel(), nm.device
    hit = torch.empty(R, dtype=torch.uint8, device=dev)
    vs = torch.empty(B + 1, dtype=torch.int32, device=dev)
    n_pos = torch.empty((), dtype=torch.int64, device=dev)
    check(lib().mvsdf_loss_prep(ptr(nm), ptr(om), ptr(tm), R, B, ptr(hit), ptr(vs), ptr(n_pos), stream_of(nm)), 'mvsdf_loss_prep')
    return hit.view(torch.bool), vs, n_pos


def loss_scale(gs, weights, unit_grads):
    """gs: the six scalar upstream gradients (tensors or None).  unit_grads: [d_rgb, d_grad, d_eo, d_sf] (None allowed) -> (scaled copies in
    the same order, coef_feat [1]).  weights = (w_rgb, w_eik, w_surf, w_feat, w_depth) as in loss_terms."""
    gs = [_f32(g.reshape(1)) if g is not None else None for g in gs]
    dev = next(g for g in gs if g is not None).device
    outs = [torch.empty_like(t) if t is not None else None for t in unit_grads]
    coef = torch.empty(1, dtype=torch.float32, device=dev)
    a = []
    for t, o in zip(unit_grads, outs):
        a += [t.data_ptr() if t is not None else None, o.data_ptr() if o is not None else None, t.numel() if t is not None else 0]
    w = [float(v) for v in weights[:5]]
    check(lib().mvsdf_loss_scale(_ptr_array(gs), w[0], w[1], w[2], w[3], w[4], *a, coef.data_ptr(), stream_of(coef).value), 'mvsdf_loss_scale')
    return outs, coef


# ---- RayTracing.forward with an opaque Python `sdf` callable (csrc/trace.hip, k_gen_*)
def trace_generic(sdf, cam_loc, ray_dirs, object_mask, params, training, intervals, minsdf_steps=None, chunk=100000):
    """-> (points[R,3], mask[R] bool, dists[R], counters[16]).  Every decision of the tracer runs in HIP kernels; this loop only carries
    the requested points to the callable and its values back (one host round trip per evaluation: the price of an opaque `sdf`).
    Rows reach the callable in ray order, start side before end side, in chunks of at most `chunk` rows (ray_tracing.py:217,300)."""
    cam_loc, ray_dirs = _f32(cam_loc), _f32(ray_dirs)
    B, P = ray_dirs.shape[:2]
    R, dev = B * P, ray_dirs.device
    om = object_mask.reshape(-1).contiguous()
    om = om.view(torch.uint8) if om.dtype == torch.bool else om.to(torch.uint8)
    tp = TraceParams(*params)
    n = tp.n_steps
    L, st = lib(), stream_of(ray_dirs)
    state = torch.empty(L.mvsdf_tracegen_state_bytes(R), dtype=torch.uint8, device=dev)
    req = torch.empty(R, 2, dtype=torch.uint8, device=dev)
    rpts = torch.empty(R, 2, 3, dtype=torch.float32, device=dev)
    vals = torch.zeros(R, 2, dtype=torch.float32, device=dev)
    counters = torch.empty(16, dtype=torch.int64, device=dev)
    pts = torch.empty(R, 3, dtype=torch.float32, device=dev)
    mask = torch.empty(R, dtype=torch.uint8, device=dev)
    dists = torch.empty(R, dtype=torch.float32, device=dev)
    wsb = L.mvsdf_trace_workspace_bytes_n(R, n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    iv = _f32(intervals)
    steps = _f32(minsdf_steps) if minsdf_steps is not None else None
    geo = (ptr(cam_loc), ptr(ray_dirs))

    def call(x):                                                  # the callable on [m,3] rows, chunked like the reference
        out = [sdf(c).reshape(-1).to(torch.float32) for c in torch.split(x, chunk, dim=0)]
        return (out[0] if len(out) == 1 else torch.cat(out)).contiguous()

    check(L.mvsdf_tracegen_init(C.byref(tp), *geo, ptr(om), B, P, ptr(state), ptr(req), ptr(rpts), ptr(counters), st), 'mvsdf_tracegen_init')
    flat_req, flat_pts, flat_vals = req.view(-1), rpts.view(-1, 3), vals.view(-1)
    while True:
        # rows in the reference's order: all requesting start sides (ray order), then all requesting end sides
        idx_s = torch.nonzero(req[:, 0]).flatten()                # host sync: the callable needs a concrete shape anyway
        idx_e = torch.nonzero(req[:, 1]).flatten()
        if idx_s.numel() + idx_e.numel() == 0:
            break
        idx = torch.cat([idx_s * 2, idx_e * 2 + 1])
        flat_vals[idx] = call(flat_pts[idx])
        check(L.mvsdf_tracegen_step(C.byref(tp), *geo, B, P, ptr(state), ptr(vals), ptr(req), ptr(rpts), ptr(counters), st), 'mvsdf_tracegen_step')
    check(L.mvsdf_tracegen_finish(C.byref(tp), *geo, B, P, 1 if training else 0, ptr(state), ptr(pts), ptr(mask), ptr(dists), ptr(counters),
                                  ptr(ws), C.c_size_t(wsb), st), 'mvsdf_tracegen_finish')
    cnt = counters.tolist()
    n_s, n_m = cnt[5], cnt[6]                                     # MVSDF_CNT_N_SAMPLER, MVSDF_CNT_N_MINSDF
    marks = torch.empty(R, dtype=torch.uint8, device=dev)
    if n_s > 0:                                                   # ray_sampler, ray_tracing.py:198-239
        rows = torch.empty(n_s * n, 3, dtype=torch.float32, device=dev)
        check(L.mvsdf_tracegen_rows(C.byref(tp), 0, *geo, B, P, ptr(iv), n_s, ptr(ws), ptr(rows), st), 'mvsdf_tracegen_rows')
        sv = call(rows)
        check(L.mvsdf_tracegen_reduce(C.byref(tp), 0, *geo, B, P, 1 if training else 0, ptr(iv), ptr(steps), ptr(sv), ptr(pts), ptr(mask),
                                      ptr(dists), ptr(counters), ptr(ws), ptr(marks), st), 'mvsdf_tracegen_reduce')
        n_sec = int(counters[4])                                  # MVSDF_CNT_N_SECANT
        if n_sec > 0:                                             # secant, ray_tracing.py:241-256
            sp = torch.empty(n_sec, 3, dtype=torch.float32, device=dev)
            sec = lambda op, v=None: check(L.mvsdf_tracegen_secant(C.byref(tp), op, *geo, B, P, n_sec, ptr(v), ptr(sp), ptr(pts), ptr(dists),
                                                                   ptr(counters), ptr(ws), st), 'mvsdf_tracegen_secant')
            for _ in range(tp.n_secant):
                sec(0)
                sec(1, call(sp))
            sec(2)
            counters[2] += n_sec * tp.n_secant                    # MVSDF_CNT_ROWS_SECANT
    if training and n_m > 0:                                      # minimal_sdf_points, ray_tracing.py:280-308
        assert steps is not None, 'training needs minsdf_steps'
        rows = torch.empty(n_m * n, 3, dtype=torch.float32, device=dev)
        check(L.mvsdf_tracegen_rows(C.byref(tp), 1, *geo, B, P, ptr(steps), n_m, ptr(ws), ptr(rows), st), 'mvsdf_tracegen_rows')
        sv = call(rows)
        check(L.mvsdf_tracegen_reduce(C.byref(tp), 1, *geo, B, P, 1, ptr(iv), ptr(steps), ptr(sv), ptr(pts), ptr(mask), ptr(dists), ptr(counters),
                                      ptr(ws), None, st), 'mvsdf_tracegen_reduce')
    return pts, mask.view(torch.bool), dists, counters
