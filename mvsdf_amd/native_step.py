"""Python side of the native step driver (csrc/step_driver.hip, include/mvsdf_hip.h "native step driver").

A training step used to be ~45 ctypes calls plus autograd glue issued from Python (1.6 ms of host time).  Here IDRNetwork.forward in training
mode is ONE C call + the one host wait for the hit counts, `loss.backward()` through it is ONE C call, and IDRLoss.forward / its backward are
one C call each; the interpreter only wraps the regions of the forward block as tensors and routes them through two autograd nodes.

Memory: one `fwd` block per forward (outputs of the reference's dict + everything the backward reads), allocated from torch's caching
allocator, so holding on to a step's outputs or running two forwards before a backward behaves like it does with the reference; the backward's
scratch block is kept per shape."""
import ctypes as C
import operator

import torch

from ._lib import TraceParams, check, lib

STEP_MAX_LAYERS = 24
_DATA_PTR = torch.Tensor.data_ptr
_VERSION_OF = operator.attrgetter('_version')


class StepDesc(C.Structure):
    _fields_ = [('B', C.c_int), ('P', C.c_int), ('n_eik', C.c_int), ('n_ds', C.c_int), ('n_sdf', C.c_int), ('n_render', C.c_int),
                ('N', C.c_int * STEP_MAX_LAYERS), ('K', C.c_int * STEP_MAX_LAYERS), ('skip_mask', C.c_uint), ('multires', C.c_int),
                ('view_spec', C.c_int), ('trace_dtype', C.c_int), ('use_object_mask', C.c_int), ('tp', TraceParams), ('mt', C.c_int),
                ('mt_samples', C.c_int)]


class StepParams(C.Structure):
    _fields_ = [('v', C.c_void_p * STEP_MAX_LAYERS), ('g', C.c_void_p * STEP_MAX_LAYERS), ('b', C.c_void_p * STEP_MAX_LAYERS)]


class StepInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('uv', 'pose', 'intrinsics', 'object_mask', 'object_mask_true', 'intervals', 'minsdf_steps', 'eik_points',
                                          'ds_on', 'ds_jit', 'ds_counts', 'host_stage')]


class StepLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ('fwd_bytes', 'bwd_bytes', 'ray_dirs', 'cam_loc', 'points', 'mask', 'dists', 'counters', 'object_mask_out',
                                          'rgb_values', 'sdf_output', 'diff_pts', 'eik_out', 'points_hom', 'grad_theta', 'surf', 'perm', 'dflat',
                                          'dflat_floats')]


class LossArgs(C.Structure):
    _fields_ = [('R', C.c_int), ('B', C.c_int), ('N', C.c_int), ('n_grad', C.c_int), ('n_depth', C.c_int), ('n_surf', C.c_int),
                ('net_mask', C.c_void_p), ('obj_mask', C.c_void_p), ('true_mask', C.c_void_p), ('rgb', C.c_void_p), ('rgb_gt', C.c_void_p),
                ('grad_theta', C.c_void_p), ('eik_out', C.c_void_p), ('surf', C.c_void_p), ('diff_pts', C.c_void_p), ('points_hom', C.c_void_p),
                ('feat_on', C.c_int), ('surf_on', C.c_int), ('V', C.c_int), ('C', C.c_int), ('H', C.c_int), ('W', C.c_int),
                ('feat', C.c_void_p), ('feat_strides', C.c_longlong * 4), ('feat_src', C.c_void_p), ('src_strides', C.c_longlong * 5),
                ('cam', C.c_void_p), ('src_cams', C.c_void_p), ('size', C.c_void_p), ('center', C.c_void_p),
                ('depths', C.c_void_p), ('dB', C.c_int), ('dh', C.c_int), ('dw', C.c_int), ('depth_cams', C.c_void_p),
                ('out_thresh_perc', C.c_float), ('far_thresh', C.c_float), ('far_att', C.c_float), ('near_thresh', C.c_float), ('near_att', C.c_float),
                ('w_rgb', C.c_float), ('w_eik', C.c_float), ('w_surf', C.c_float), ('w_feat', C.c_float), ('w_depth', C.c_float),
                ('use_invalid', C.c_int), ('smooth', C.c_float), ('inv_counts', C.c_void_p),
                # deferred step: device pointer to {N, n_true}; N / n_grad / n_depth / n_surf above are then upper bounds (N = R), see include/mvsdf_hip.h
                ('counts_dev', C.c_void_p), ('n_eik', C.c_int), ('n_ds', C.c_int), ('d_mask', C.c_int), ('e_mask', C.c_int)]


class LossLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ('bytes', 'out', 'hit', 'view_start', 'n_pos', 'loss_pp', 'dpts', 'dist_r', 'weight', 'd_rgb', 'd_grad',
                                          'd_eo', 'd_sf', 's_rgb', 's_grad', 's_eo', 's_sf', 's_diff')]


_bound = False


def _bind():
    global _bound
    L = lib()
    if not _bound:
        L.mvsdf_step_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mvsdf_step_destroy.argtypes = [C.c_void_p]
        L.mvsdf_step_destroy.restype = None
        L.mvsdf_step_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.mvsdf_step_wait_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.mvsdf_step_seq.argtypes = [C.c_void_p]
        L.mvsdf_step_seq.restype = C.c_longlong
        L.mvsdf_step_counts_offset.argtypes = [C.c_void_p]
        L.mvsdf_step_counts_offset.restype = C.c_size_t
        L.mvsdf_step_wait_counts_seq.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p]
        L.mvsdf_step_done_seq.argtypes = [C.c_void_p, C.c_void_p]
        L.mvsdf_step_done_seq.restype = C.c_longlong
        L.mvsdf_step_can_defer.argtypes = [C.c_void_p]
        L.mvsdf_step_saved_offsets.argtypes = [C.c_void_p, C.c_void_p]
        L.mvsdf_step_backward.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 10 + [C.c_int, C.c_void_p]
        L.mvsdf_step_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.mvsdf_step_trace_times.argtypes = [C.c_void_p, C.c_void_p]
        L.mvsdf_step_times.argtypes = [C.c_void_p, C.c_void_p]
        L.mvsdf_loss_layout.argtypes = [C.c_void_p, C.c_void_p]
        L.mvsdf_loss_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mvsdf_loss_backward.argtypes = [C.c_void_p] * 9
        _bound = True
    return L


loss_timing = None                                             # bench.py sets a list: (start, end) events around every mvsdf_loss_forward are appended


def _timed_loss_forward(L, args, blk, dev):
    if loss_timing is None:
        check(L.mvsdf_loss_forward(C.byref(args), blk.data_ptr(), _stream(dev)), 'mvsdf_loss_forward')
        return
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    check(L.mvsdf_loss_forward(C.byref(args), blk.data_ptr(), _stream(dev)), 'mvsdf_loss_forward')
    ev[1].record()
    loss_timing.append(ev)


def _region(block, off, nbytes, dtype, shape):
    """A typed view of `nbytes` bytes of a uint8 block at byte offset `off` (offsets are multiples of 256)."""
    return block[off:off + nbytes].view(dtype).view(shape)


def _strides(shape):
    st, acc = [], 1
    for n in reversed(shape):
        st.append(acc)
        acc *= max(int(n), 1)
    return tuple(reversed(st))


class Block:
    """A forward block with its float32 alias: regions are cut with ONE as_strided each (the three-op slice / view / view of `_region`
    costs ~3 us per tensor, a dozen of them per step)."""
    __slots__ = ('u8', 'f32')

    def __init__(self, nbytes, device):
        self.u8 = torch.empty((nbytes + 3) // 4 * 4, dtype=torch.uint8, device=device)
        self.f32 = self.u8.view(torch.float32)

    def f(self, off, shape):
        return torch.as_strided(self.f32, shape, _strides(shape), off >> 2)

    def b(self, off, shape):
        return torch.as_strided(self.u8, shape, _strides(shape), off)

    def data_ptr(self):
        return self.u8.data_ptr()


def _stream(dev):
    """torch's CURRENT stream of `dev` as the raw hipStream_t (the private accessor torch.cuda.current_stream is built on: a fifth of its host time)."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()))


class NativeStep:
    """Host-side state of one step shape (mvsdf_step_create): layout, pinned count buffer, events; plus the cached ctypes views of the
    parameters and the backward's scratch block."""

    def __init__(self, desc, device):
        L = _bind()
        self.desc, self.device = desc, device
        self.layout = StepLayout()
        h = C.c_void_p()
        check(L.mvsdf_step_create(C.byref(desc), C.byref(self.layout), C.byref(h)), 'mvsdf_step_create')
        self._h = h
        self.R, self.E = desc.B * desc.P, desc.n_eik + 2 * desc.n_ds
        self.Nout = desc.N[desc.n_sdf - 1]
        self.nl = desc.n_sdf + desc.n_render
        self._bwd = None
        self._prm_key, self._prm, self._prm_list = None, None, None
        self._grad_key, self._grad_arrays = None, None
        self._counts = (C.c_longlong * 4)()
        self._counts_peek = (C.c_longlong * 4)()
        self.inputs = StepInputs()
        self.timing = False
        self.can_defer = bool(L.mvsdf_step_can_defer(h))          # every launch of the backward has a device-count form (the deferred step)
        self.counts_off = int(L.mvsdf_step_counts_offset(h))      # int64 counts[4] inside a forward block

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                lib().mvsdf_step_destroy(h)
            except Exception:
                pass

    # ---- parameters
    def params(self, vs, gs, bs):
        """ctypes struct of the raw parameter pointers, rebuilt only when a storage moved."""
        if self._prm_list is None or len(self._prm_list) != len(vs) + len(bs) + sum(1 for g in gs if g is not None):
            self._prm_list = list(vs) + [g for g in gs if g is not None] + list(bs)
        key = tuple(map(_DATA_PTR, self._prm_list))               # (a storage that moved -- .to(), FlatAdam taking the parameters over -- shows here)
        if key != self._prm_key:
            for p in list(vs) + [g for g in gs if g is not None] + list(bs):
                assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous(), 'float32 contiguous parameters on the GPU expected'
            prm = StepParams()
            for l, (v, g, b) in enumerate(zip(vs, gs, bs)):
                prm.v[l], prm.g[l], prm.b[l] = v.data_ptr(), (g.data_ptr() if g is not None else None), b.data_ptr()
            self._prm_key, self._prm = key, prm
        return self._prm

    def grad_arrays(self, vs, gs, bs):
        """Pointer arrays of the parameters' .grad buffers (the gradient sink), or None when one is missing / not a plain fp32 buffer."""
        ps = self._prm_list if self._prm_list is not None else list(vs) + [g for g in gs if g is not None] + list(bs)
        grads = [p.grad for p in ps]
        for g_ in grads:                                          # (not `None in grads`: `in` compares with ==, i.e. 42 tensor comparisons)
            if g_ is None:
                return None
        key = tuple(map(_DATA_PTR, grads))
        if key != self._grad_key:
            if not all(p.grad.is_cuda and p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in ps):
                return None
            n = STEP_MAX_LAYERS
            dv, dg, db = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
            for l, (v, g, b) in enumerate(zip(vs, gs, bs)):
                dv[l], dg[l], db[l] = v.grad.data_ptr(), (g.grad.data_ptr() if g is not None else None), b.grad.data_ptr()
            self._grad_key, self._grad_arrays = key, (dv, dg, db)
        return self._grad_arrays

    # ---- calls
    def forward(self, prm, d_mask, e_mask):
        """-> the forward block (enqueued, nothing waited for); self.seq() names this forward."""
        fwd = Block(self.layout.fwd_bytes, self.device)
        check(lib().mvsdf_step_forward(self._h, C.byref(prm), C.byref(self.inputs), d_mask, e_mask, fwd.data_ptr(), _stream(self.device)),
              'mvsdf_step_forward')
        return fwd

    def wait_counts(self):
        check(lib().mvsdf_step_wait_counts(self._h, self._counts), 'mvsdf_step_wait_counts')
        return tuple(self._counts)

    def seq(self):
        """Sequence number of the last forward of this step object (the deferred step keeps it to ask for that forward's counts later)."""
        return int(lib().mvsdf_step_seq(self._h))

    def wait_counts_seq(self, seq, fwd):
        """The counts of forward `seq` (blocks until its ray partition has run).  A record older than the pinned ring (64 forwards) is read from the
        forward block itself after a stream synchronisation."""
        rc = lib().mvsdf_step_wait_counts_seq(self._h, seq, self._counts)
        if rc == -4:
            torch.cuda.current_stream(self.device).synchronize()
            return tuple(int(v) for v in fwd.u8[self.counts_off:self.counts_off + 32].view(torch.int64).cpu())
        check(rc, 'mvsdf_step_wait_counts_seq')
        return tuple(self._counts)

    def partition_done(self, seq):
        """Blocks until the ray partition of forward `seq` has run (everything enqueued before it -- the prologue reading the pinned draws -- is then done)."""
        rc = lib().mvsdf_step_wait_counts_seq(self._h, seq, self._counts)
        if rc != -4:                                              # (-4: overwritten by a forward 64 steps later, i.e. long done)
            check(rc, 'mvsdf_step_wait_counts_seq')

    def saved_offsets(self):
        """Byte offsets inside a forward block of {x_eval, y_eval, n_eval, view_sorted, render_ctx, rgb_sorted} (mvsdf_step_saved_offsets): inspection only."""
        out = (C.c_size_t * 6)()
        check(lib().mvsdf_step_saved_offsets(self._h, out), 'mvsdf_step_saved_offsets')
        return dict(zip(('x_eval', 'y_eval', 'n_eval', 'view_sorted', 'render_ctx', 'rgb_sorted'), (int(v) for v in out)))

    def hint_N(self):
        """N of the newest forward whose counts have arrived (no waiting), or -1: only selects kernel forms of a deferred backward."""
        return int(self._counts_peek[0]) if lib().mvsdf_step_done_seq(self._h, self._counts_peek) > 0 else -1


    def bwd_block(self):
        if self._bwd is None:
            self._bwd = torch.empty(self.layout.bwd_bytes, dtype=torch.uint8, device=self.device)
        return self._bwd

    def backward(self, prm, N, n_true, d_mask, e_mask, use_geo, ups, fwd, targets, accumulate):
        # (ups: tensors, or raw device addresses as ints -- the deferred direct route hands over regions of the loss block without wrapping them)
        p = [None if t is None else C.c_void_p(t if isinstance(t, int) else t.data_ptr()) for t in ups]
        dv, dg, db = targets
        check(lib().mvsdf_step_backward(self._h, C.byref(prm), N, n_true, d_mask, e_mask, 1 if use_geo else 0, p[0], p[1], p[2], p[3], p[4],
                                        fwd.data_ptr(), self.bwd_block().data_ptr(), dv, dg, db, 1 if accumulate else 0, _stream(self.device)),
              'mvsdf_step_backward')

    def set_timing(self, on):
        check(lib().mvsdf_step_set_timing(self._h, 1 if on else 0), 'mvsdf_step_set_timing')
        self.timing = bool(on)

    def trace_times(self):
        """-> (sphere tracing, sampler rows, secant + min-sdf rows) in ms of the last forward; the stream must have been synchronised."""
        ms = (C.c_float * 3)()
        check(lib().mvsdf_step_trace_times(self._h, ms), 'mvsdf_step_trace_times')
        return tuple(ms)

    def times(self):
        """-> (sphere tracing, sampler rows, secant + min-sdf rows, forward behind the tracer, backward, whole forward) in ms of the last step; the
        stream must have been synchronised."""
        ms = (C.c_float * 6)()
        check(lib().mvsdf_step_times(self._h, ms), 'mvsdf_step_times')
        return tuple(ms)


class StepRecord:
    """What one forward leaves behind for its backward and for the output dict."""
    __slots__ = ('step', 'fwd', 'prm', 'params', 'N', 'n_true', 'counts', 'd_mask', 'e_mask', 'use_geo', 'n_layers', 'vs', 'gs', 'bs', 'keep', 'done', 'versions',
                 'seq', 'inputs_keep', 'live', '__weakref__')

    def __init__(self):
        self.fwd = self.N = self.n_true = self.counts = self.seq = None
        self.done = False

    def resolve(self):
        """(N, n_true): host-side counts of this forward -- the classic step asks right after its forward, a deferred step only when somebody reads an
        N-shaped output (its loss / backward / optimiser never do)."""
        if self.N is None:
            self.counts = self.step.wait_counts_seq(self.seq, self.fwd)
            self.N, self.n_true = int(self.counts[0]), int(self.counts[1])
        return self.N, self.n_true

    def group_rows(self, N):
        """(nd, ne): rows of eikonal_output / grad_theta for N hit rows (point groups [hit | eikonal | on-surface | jittered], idr.py:253-286)."""
        d = self.step.desc
        sizes = (N, d.n_eik, d.n_ds, d.n_ds)
        return (sum(c for g, c in enumerate(sizes) if self.d_mask >> g & 1), sum(c for g, c in enumerate(sizes) if self.e_mask >> g & 1))


class _NativeStepFn(torch.autograd.Function):
    """IDRNetwork.forward (training) as one autograd node over the raw parameters: forward = mvsdf_step_forward + the one host wait,
    backward = mvsdf_step_backward (rendering-net backward, SampleNetwork's scalar, the first/second-order SDF backward, weight gradients,
    weight-norm fold backward).  Inside functional.grad_sink() with FlatAdam-style persistent .grad buffers the gradients are ADDED
    into those buffers by the last launch and autograd sees None."""

    @staticmethod
    def forward(ctx, rec, *params):
        st = rec.step
        if rec.fwd is None:                                       # (a deferred step that is being materialised has run its forward already)
            enqueue_forward(rec)
        N, n_true = rec.resolve()                                 # the one host wait of the classic training forward
        L, d, R, E, f = st.layout, st.desc, st.R, st.E, rec.fwd
        nd, ne = rec.group_rows(N)
        diff = f.f(L.diff_pts, (N, 3))
        rgb = f.f(L.rgb_values, (R, 3))
        gth = f.f(L.grad_theta, (ne, 3))
        eo = f.f(L.eik_out, (1, nd))
        surf = f.f(L.surf, (n_true + d.n_eik,))
        rec.keep = (nd, ne)
        ctx.rec = rec
        return diff, rgb, gth, eo, surf

    @staticmethod
    def backward(ctx, d_diff, d_rgb, d_gth, d_eo, d_si):
        from .functional import grad_sink, mark_sink_written
        rec = ctx.rec
        _check_backward_allowed(rec)
        st = rec.step
        ups = [None if t is None else (t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float()) for t in (d_diff, d_rgb, d_gth, d_eo, d_si)]
        vs, gs, bs = rec.vs, rec.gs, rec.bs
        n = len(vs)
        sink = None
        if grad_sink.depth > 0 and all(getattr(p, '_mv_grad_sink', False) and p.requires_grad for p in rec.params if p is not None):
            sink = st.grad_arrays(vs, gs, bs)
        if sink is not None:
            st.backward(rec.prm, rec.N, rec.n_true, rec.d_mask, rec.e_mask, rec.use_geo, ups, rec.fwd, sink, True)
            mark_sink_written(vs)
            return (None,) * (1 + 3 * n)
        # ordinary autograd route: fresh gradient tensors for every parameter (one allocation, carved into views)
        sizes = [v.numel() for v in vs] + [0 if g is None else g.numel() for g in gs] + [b.numel() for b in bs]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=st.device)
        parts = list(torch.split(flat, sizes))
        dvs = [t.view_as(v) for t, v in zip(parts[:n], vs)]
        dgs = [None if g is None else t.view_as(g) for t, g in zip(parts[n:2 * n], gs)]
        dbs = [t.view_as(b) for t, b in zip(parts[2 * n:], bs)]
        m = STEP_MAX_LAYERS
        dv, dg, db = (C.c_void_p * m)(), (C.c_void_p * m)(), (C.c_void_p * m)()
        for l in range(n):
            dv[l], dg[l], db[l] = dvs[l].data_ptr(), (dgs[l].data_ptr() if dgs[l] is not None else None), dbs[l].data_ptr()
        st.backward(rec.prm, rec.N, rec.n_true, rec.d_mask, rec.e_mask, rec.use_geo, ups, rec.fwd, (dv, dg, db), False)
        return (None,) + tuple(dvs) + tuple(dgs) + tuple(dbs)


def enqueue_forward(rec):
    """mvsdf_step_forward for this record: everything of IDRNetwork.forward is on the stream when this returns; nothing is waited for."""
    st = rec.step
    rec.fwd = st.forward(rec.prm, rec.d_mask, rec.e_mask)
    rec.seq = st.seq()
    # the backward reads the parameters again (weight-norm fold backward over prm->v / g): an in-place update between this forward and its backward
    # (forward A, forward B, backward B, opt.step(), backward A) must raise like autograd's saved-tensor check does, not mix old activations with new weights
    rec.versions = tuple(map(_VERSION_OF, rec.live))


def _check_backward_allowed(rec):
    if getattr(rec, 'done', False):
        raise RuntimeError('the backward of this step already ran through FlatAdam.backward (its buffers are released after one backward, '
                           'like autograd\'s)')
    if tuple(map(_VERSION_OF, rec.live)) != rec.versions:
        raise RuntimeError('one of the variables needed for gradient computation has been modified by an inplace operation: a parameter of the '
                           'network changed between this step\'s forward and its backward (e.g. optimizer.step() in between)')


def run_step(rec):
    """-> (diff_surf_pts, rgb_values, grad_theta, eikonal_output, surf_indicator_output) linked to autograd; rec.fwd / N / n_true set."""
    return _NativeStepFn.apply(rec, *rec.params)


# ------------------------------------------------------------------------------------------------------------------------------
class _NativeLossFn(torch.autograd.Function):
    """IDRLoss.forward (loss.py:176-219) as one node: forward = mvsdf_loss_forward (mask bookkeeping, feature consistency, depth carving,
    every term + its unit gradient: 4 launches), backward = mvsdf_loss_backward (one launch)."""

    @staticmethod
    def forward(ctx, args, keep, rgb, grad_theta, eik_out, surf, diff_pts):
        L = _bind()
        lo = LossLayout()
        check(L.mvsdf_loss_layout(C.byref(args), C.byref(lo)), 'mvsdf_loss_layout')
        dev = rgb.device
        blk = torch.empty(lo.bytes, dtype=torch.uint8, device=dev)
        _timed_loss_forward(L, args, blk, dev)
        ctx.args, ctx.blk, ctx.keep, ctx.lo = args, blk, keep, lo
        ctx.ins = (rgb, grad_theta, eik_out, surf, diff_pts)     # (for direct_backward: which node produced them)
        ctx.shapes = (rgb.shape, grad_theta.shape if grad_theta is not None else None, eik_out.shape, surf.shape if surf is not None else None,
                      diff_pts.shape)
        ctx.set_materialize_grads(False)                         # unused scalars arrive as None, not as zero tensors
        out = _region(blk, lo.out, 24, torch.float32, (6,))
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        a, dev = ctx.args, ctx.blk.device
        garr = (C.c_void_p * 6)()
        keep = []
        for k, g in enumerate(gs):
            if g is not None:
                g = g.reshape(1).float().contiguous()
                keep.append(g)
                garr[k] = g.data_ptr()
        f = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
        feat = bool(a.feat_on) and a.N > 0
        sizes = [a.R * 3, a.n_grad * 3, a.n_depth, a.n_surf if a.surf_on else 0, a.N * 3 if feat else 0]
        flat = f(sum(sizes))
        g_rgb, g_grad, g_eo, g_sf, g_diff = torch.split(flat, sizes)
        p = lambda t, on=True: C.c_void_p(t.data_ptr()) if (on and t.numel() > 0) else None
        check(lib().mvsdf_loss_backward(C.byref(a), ctx.blk.data_ptr(), garr, p(g_rgb), p(g_grad), p(g_eo), p(g_sf, bool(a.surf_on)),
                                        p(g_diff, feat), _stream(dev)), 'mvsdf_loss_backward')
        sh = ctx.shapes
        return (None, None, g_rgb.view(sh[0]), g_grad.view(sh[1]) if (sh[1] is not None and a.n_grad > 0) else None, g_eo.view(sh[2]),
                g_sf.view(sh[3]) if (sh[3] is not None and a.surf_on and a.n_surf > 0) else None, g_diff.view(sh[4]) if feat else None)


def loss_forward(args, keep, rgb, grad_theta, eik_out, surf, diff_pts):
    return _NativeLossFn.apply(args, keep, rgb, grad_theta, eik_out, surf, diff_pts)


# ------------------------------------------------------------------------------------------------------------------------------
class _DeferredStepLossFn(torch.autograd.Function):
    """IDRLoss.forward on the outputs of a DEFERRED step (IDRNetwork.forward returned before the host knew the hit counts) as ONE node over the raw
    parameters: forward = mvsdf_loss_forward with the counts read on the device (MvsdfLossArgs.counts_dev), backward = mvsdf_loss_backward +
    mvsdf_step_backward(N < 0).  No host wait anywhere: a loop `forward -> loss -> backward -> optimiser` enqueues whole steps ahead of the GPU.
    Same kernels and the same bits as the classic two-node graph (tests/test_gpu_deferred.py)."""

    @staticmethod
    def forward(ctx, rec, args, keep, *params):
        L = _bind()
        lo = LossLayout()
        check(L.mvsdf_loss_layout(C.byref(args), C.byref(lo)), 'mvsdf_loss_layout')
        st = rec.step
        dev = st.device
        blk = torch.empty(lo.bytes, dtype=torch.uint8, device=dev)
        _timed_loss_forward(L, args, blk, dev)
        ctx.rec, ctx.args, ctx.blk, ctx.keep, ctx.lo = rec, args, blk, keep, lo
        ctx.set_materialize_grads(False)
        out = _region(blk, lo.out, 24, torch.float32, (6,))
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        rec, a, lo, blk = ctx.rec, ctx.args, ctx.lo, ctx.blk
        dev = blk.device
        garr = (C.c_void_p * 6)()
        keep = []
        for k, g in enumerate(gs):
            if g is not None:
                g = g.reshape(1).float().contiguous()
                keep.append(g)
                garr[k] = g.data_ptr()
        feat = bool(a.feat_on) and a.N > 0
        sizes = [a.R * 3, a.n_grad * 3, a.n_depth, a.n_surf if a.surf_on else 0, a.N * 3 if feat else 0]      # upper bounds: the valid rows lead
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        g_rgb, g_grad, g_eo, g_sf, g_diff = torch.split(flat, sizes)
        p = lambda t, on=True: C.c_void_p(t.data_ptr()) if (on and t.numel() > 0) else None
        check(lib().mvsdf_loss_backward(C.byref(a), blk.data_ptr(), garr, p(g_rgb), p(g_grad), p(g_eo), p(g_sf, bool(a.surf_on)),
                                        p(g_diff, feat), _stream(dev)), 'mvsdf_loss_backward')
        ups = (g_diff if feat else None, g_rgb, g_grad if a.n_grad > 0 else None, g_eo if a.n_depth > 0 else None,
               g_sf if (a.surf_on and a.n_surf > 0) else None)
        return (None, None, None) + _deferred_step_backward(rec, ups)


def _sink_of(rec):
    """The pointer arrays of the parameters' persistent .grad buffers when the gradient sink is on for ALL of them (functional.grad_sink), else None."""
    from .functional import grad_sink
    if grad_sink.depth <= 0:
        return None
    for p in rec.live:
        if not (getattr(p, '_mv_grad_sink', False) and p.requires_grad):
            return None
    return rec.step.grad_arrays(rec.vs, rec.gs, rec.bs)


def _deferred_step_backward(rec, ups, sink=False):
    """mvsdf_step_backward with device-side counts for a deferred step; -> the gradients of rec.params (None each when they went into the sink).
    sink: what _sink_of(rec) returned to a caller that already asked (False: ask here)."""
    from .functional import mark_sink_written
    _check_backward_allowed(rec)
    st = rec.step
    vs, gs, bs = rec.vs, rec.gs, rec.bs
    n = len(vs)
    hint = rec.N if rec.N is not None else st.hint_N()
    if sink is False:
        sink = _sink_of(rec)
    if sink is not None:
        st.backward(rec.prm, -1, hint, rec.d_mask, rec.e_mask, rec.use_geo, ups, rec.fwd, sink, True)
        mark_sink_written(vs)
        return (None,) * (3 * n)
    sizes = [v.numel() for v in vs] + [0 if g is None else g.numel() for g in gs] + [b.numel() for b in bs]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=st.device)
    parts = list(torch.split(flat, sizes))
    dvs = [t.view_as(v) for t, v in zip(parts[:n], vs)]
    dgs = [None if g is None else t.view_as(g) for t, g in zip(parts[n:2 * n], gs)]
    dbs = [t.view_as(b) for t, b in zip(parts[2 * n:], bs)]
    m = STEP_MAX_LAYERS
    dv, dg, db = (C.c_void_p * m)(), (C.c_void_p * m)(), (C.c_void_p * m)()
    for l in range(n):
        dv[l], dg[l], db[l] = dvs[l].data_ptr(), (dgs[l].data_ptr() if dgs[l] is not None else None), dbs[l].data_ptr()
    st.backward(rec.prm, -1, hint, rec.d_mask, rec.e_mask, rec.use_geo, ups, rec.fwd, (dv, dg, db), False)
    return tuple(dvs) + tuple(dgs) + tuple(dbs)


def deferred_loss_forward(rec, args, keep):
    """The six loss scalars of a deferred step (see _DeferredStepLossFn)."""
    return _DeferredStepLossFn.apply(rec, args, keep, *rec.params)


_one = {}


def direct_backward(loss):
    """`loss.backward()` without the autograd engine, for the graph the native step builds: loss = output k of a _NativeLossFn node whose five
    inputs are exactly the five outputs of ONE _NativeStepFn node whose parameters all carry a gradient sink (optim.FlatAdam).  The two
    backward functions are then called in line on the calling thread (the engine's hand-over to its device thread and back costs ~0.2 ms per
    step, more than both C calls together).  -> True when it ran; False when the graph is anything else (the caller falls back to
    loss.backward(): same numbers)."""
    from .functional import grad_sink
    node = loss.grad_fn
    if node is not None and isinstance(getattr(node, 'rec', None), StepRecord) and hasattr(node, 'lo') and grad_sink.depth > 0:
        return _direct_backward_deferred(loss, node)
    if node is None or getattr(node, 'ins', None) is None or not hasattr(node, 'args') or grad_sink.depth <= 0:
        return False
    rgb, gth, eo, sf, pts = node.ins
    snode = rgb.grad_fn
    rec = getattr(snode, 'rec', None)
    if rec is None or getattr(rec, 'done', False):
        return False
    for k, t in enumerate((pts, rgb, gth, eo, sf)):              # the step node's outputs in its order, nothing else in between
        if t is None or t.grad_fn is not snode or t.output_nr != k:
            return False
        if t._backward_hooks or t.retains_grad:                  # register_hook / retain_grad on a step output: only the autograd engine honours them
            return False
    if not all(getattr(p, '_mv_grad_sink', False) and p.requires_grad for p in rec.params if p is not None):
        return False
    if rec.step.grad_arrays(rec.vs, rec.gs, rec.bs) is None:
        return False
    dev = loss.device
    one = _one.get(dev)
    if one is None:
        one = _one[dev] = torch.ones((), dtype=torch.float32, device=dev)
    if loss.output_nr == 0:
        # d(total loss): mvsdf_loss_forward already left the weighted gradients in its block (bit-identical to the backward launch for an upstream of 1)
        a, lo, blk = node.args, node.lo, node.blk
        f32 = blk.view(torch.float32)
        cut = lambda off, n, shape: torch.as_strided(f32, shape, _strides(shape), off >> 2) if n > 0 else None
        feat = bool(a.feat_on) and a.N > 0
        g_rgb = cut(lo.s_rgb, a.R, (a.R, 3))
        g_grad = cut(lo.s_grad, a.n_grad, (a.n_grad, 3))
        g_eo = cut(lo.s_eo, a.n_depth, (a.n_depth,))
        g_sf = cut(lo.s_sf, a.n_surf, (a.n_surf,)) if a.surf_on else None
        g_diff = cut(lo.s_diff, a.N, (a.N, 3)) if feat else None
        _NativeStepFn.backward(snode, g_diff, g_rgb, g_grad, g_eo, g_sf)
    else:
        gs = [None] * 6
        gs[loss.output_nr] = one
        grads = _NativeLossFn.backward(node, *gs)
        _NativeStepFn.backward(snode, grads[6], grads[2], grads[3], grads[4], grads[5])
    rec.done = True
    return True


def _direct_backward_deferred(loss, node):
    """direct_backward for the one-node graph of a deferred step (_DeferredStepLossFn): still no host wait."""
    rec = node.rec
    if getattr(rec, 'done', False) or loss._backward_hooks or loss.retains_grad:
        return False
    sink = _sink_of(rec)
    if sink is None:
        return False
    if loss.output_nr == 0:
        # d(total loss): mvsdf_loss_forward already left the weighted gradients in its block (regions sized for N = R, the valid rows lead): raw addresses
        a, lo, base = node.args, node.lo, node.blk.data_ptr()
        feat = bool(a.feat_on) and a.N > 0
        ups = (base + lo.s_diff if feat else None, base + lo.s_rgb, base + lo.s_grad if a.n_grad > 0 else None,
               base + lo.s_eo if a.n_depth > 0 else None, base + lo.s_sf if (a.surf_on and a.n_surf > 0) else None)
        _deferred_step_backward(rec, ups, sink)
    else:
        dev = loss.device
        one = _one.get(dev)
        if one is None:
            one = _one[dev] = torch.ones((), dtype=torch.float32, device=dev)
        gs = [None] * 6
        gs[loss.output_nr] = one
        _DeferredStepLossFn.backward(node, *gs)
    rec.done = True
    return True
