"""ctypes front end of the C oracle (oracle/liboracle_mvsdf.so).

TEST INFRASTRUCTURE -- importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  numpy in, numpy out; no torch, no GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, 'liboracle_mvsdf.so')
    src = [os.path.join(_HERE, f) for f in ('oracle_mvsdf.c', 'det_math.h')]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle_mvsdf.so'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def bf16_round(a):
    """fp32 -> nearest-even bf16, returned as fp32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    u = (u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)
    return u.view(np.float32)


class Net:
    """Folded SDF network (weights from a reference-layout state dict)."""

    def __init__(self, state, prefix='implicit_network', skip_in=(4,), multires=6, bf16=False):
        self.multires = multires
        # bf16 twin (BASELINE configs[4]): weights rounded here, activations in C.  bf16='weights': only the weights are rounded (the twin of
        # trace_dtype 2: fp32 activations and arithmetic on bf16-rounded weights)
        # bf16='f32x3': the fp32 weights UNROUNDED, arithmetic of trace_dtype 5 (three bf16 terms per weight and activation on a bit-level model of
        # v_mfma_f32_16x16x32_bf16: oracle_mvsdf.c::sdf_row_f32x3) -- reproduces that engine bit for bit
        self.mode = 3 if bf16 == 'f32x3' else (1 if (bool(bf16) and bf16 != 'weights') else 0)
        self.bf16 = self.mode == 1
        if bf16 == 'f32x3':
            bf16 = False
        self.W, self.b = [], []
        l = 0
        while '%s.lin%d.weight_v' % (prefix, l) in state:
            v = _f(state['%s.lin%d.weight_v' % (prefix, l)])
            g = _f(state['%s.lin%d.weight_g' % (prefix, l)]).reshape(-1)
            self.W.append(bf16_round(fold(v, g)) if bf16 else fold(v, g))
            self.b.append(_f(state['%s.lin%d.bias' % (prefix, l)]))
            l += 1
        self.n_layers = l
        self.skip_in = tuple(int(v) for v in skip_in)
        self.skip_layer = self.skip_in[0] if len(self.skip_in) else -1
        self.skip_mask = sum(1 << v for v in self.skip_in)
        self.ins = np.array([w.shape[1] for w in self.W], dtype=np.int32)
        self.outs = np.array([w.shape[0] for w in self.W], dtype=np.int32)
        self.Wcat = np.concatenate([w.reshape(-1) for w in self.W])
        self.bcat = np.concatenate(self.b)

    def args(self):
        return (C.c_int(self.n_layers), _p(self.ins), _p(self.outs), C.c_int(self.skip_mask), C.c_int(self.multires),
                _p(self.Wcat), _p(self.bcat))


def fold(v, g):
    v = _f(v)
    g = _f(g).reshape(-1)
    w = np.empty_like(v)
    lib().orc_fold(_p(v), _p(g), C.c_int(v.shape[0]), C.c_int(v.shape[1]), _p(w))
    return w


def pe(x, multires):
    x = _f(x)
    out = np.empty((x.shape[0], 3 + 6 * multires), np.float32)
    lib().orc_pe(_p(x), C.c_int(x.shape[0]), C.c_int(multires), _p(out))
    return out


def sdf_forward(net, x, ncols=None):
    x = _f(x)
    ncols = int(net.outs[-1]) if ncols is None else ncols
    y = np.empty((x.shape[0], ncols), np.float32)
    lib().orc_set_bf16(C.c_int(getattr(net, 'mode', 1 if net.bf16 else 0)))
    lib().orc_sdf_forward(*net.args(), _p(x), C.c_int(x.shape[0]), C.c_int(ncols), _p(y))
    lib().orc_set_bf16(C.c_int(0))
    return y


def camera_rays(uv, pose, K):
    uv, pose, K = _f(uv), _f(pose), _f(K)
    B, P = uv.shape[:2]
    dirs = np.empty((B, P, 3), np.float32)
    cam = np.empty((B, 3), np.float32)
    lib().orc_camera_rays(_p(uv), _p(pose), _p(K), C.c_int(B), C.c_int(P), _p(dirs), _p(cam))
    return dirs, cam


def sphere_intersection(cam_loc, dirs, r=1.0):
    cam_loc, dirs = _f(cam_loc), _f(dirs)
    B, P = dirs.shape[:2]
    t = np.empty((B, P, 2), np.float32)
    m = np.empty((B, P), np.uint8)
    lib().orc_sphere_intersection(_p(cam_loc), _p(dirs), C.c_int(B), C.c_int(P), C.c_float(r), _p(t), _p(m))
    return t, m.astype(bool)


def linspace01(n):
    """torch.linspace(0, 1, n) in float32 (ray_tracing.py:206): start + i*step for the first half,
    end - (n-1-i)*step for the second (ATen RangeFactories), step = fl32(1/(n-1))."""
    step = np.float32(1.0) / np.float32(n - 1)
    i = np.arange(n)
    lo = (np.float32(0.0) + step * i.astype(np.float32)).astype(np.float32)
    hi = (np.float32(1.0) - step * (n - 1 - i).astype(np.float32)).astype(np.float32)
    return np.where(i < n // 2, lo, hi).astype(np.float32)


def trace(net, cam_loc, dirs, object_mask, training, minsdf_steps=None, intervals=None, analytic=False,
          object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=1,
          sphere_tracing_iters=10, n_steps=100, n_secant_steps=8, dist_clip=0.5, margins=False):
    """RayTracing.forward (ray_tracing.py:27-98) -> points[R,3], mask[R] bool, dists[R], rows[4]
    (+ margins[R,2] with margins=True: per ray min |sdf| and min |sdf - threshold| over every evaluation the ray made -- how far its
    sign / convergence decisions were from flipping, the quantities make_golden.py::MarginRecorder takes from the reference)."""
    cam_loc, dirs = _f(cam_loc), _f(dirs)
    B, P = dirs.shape[:2]
    R = B * P
    om = np.ascontiguousarray(np.asarray(object_mask).reshape(-1), dtype=np.uint8)
    intervals = linspace01(n_steps) if intervals is None else _f(intervals)
    steps = np.zeros(n_steps, np.float32) if minsdf_steps is None else _f(minsdf_steps)
    pts = np.empty((R, 3), np.float32)
    mask = np.empty((R,), np.uint8)
    dists = np.empty((R,), np.float32)
    rows = np.zeros(4, np.int64)
    if analytic:
        z = np.zeros(1, np.int32)
        nargs = (C.c_int(0), _p(z), _p(z), C.c_int(-1), C.c_int(0), None, None)
    else:
        nargs = net.args()
    lib().orc_set_bf16(C.c_int(getattr(net, 'mode', 0) if net is not None else 0))
    mg = np.empty((R, 2), np.float32) if margins else None
    lib().orc_trace_m(C.c_int(1 if analytic else 0), *nargs, _p(cam_loc), _p(dirs), _p(om), C.c_int(B), C.c_int(P),
                      C.c_float(object_bounding_sphere), C.c_float(sdf_threshold), C.c_float(line_search_step),
                      C.c_int(line_step_iters), C.c_int(sphere_tracing_iters), C.c_int(n_steps), C.c_int(n_secant_steps),
                      C.c_float(dist_clip), C.c_int(1 if training else 0), _p(intervals), _p(steps),
                      _p(pts), _p(mask), _p(dists), _p(rows), _p(mg) if margins else None)
    lib().orc_set_bf16(C.c_int(0))
    if margins:
        return pts, mask.astype(bool), dists, rows, mg
    return pts, mask.astype(bool), dists, rows


def _unary(name, x, nout=1):
    x = _f(x).reshape(-1)
    outs = [np.empty_like(x) for _ in range(nout)]
    getattr(lib(), name)(_p(x), C.c_int(x.size), *[_p(o) for o in outs])
    return outs[0] if nout == 1 else outs


def mfma_tiles(A, B, Cin, vector=False):
    """The oracle's model of v_mfma_f32_16x16x32_bf16 on n tiles: A [n,16,32] / B [n,16,32] bf16 bit patterns (uint16), Cin [n,16,16] fp32 -> D [n,16,16]
    (D[i][j] = Cin[i][j] + sum_k A[i][k] B[j][k] in the instruction's own order and roundings).  vector: the eight-column AVX2 form the MLP rows use."""
    A = np.ascontiguousarray(A, np.uint16); B = np.ascontiguousarray(B, np.uint16); Cin = _f(Cin)
    D = np.empty_like(Cin)
    lib().orc_mfma_tiles(_p(A), _p(B), _p(Cin), _p(D), C.c_int(A.shape[0]), C.c_int(1 if vector else 0))
    return D


def set_x3_scalar(on):
    """f32x3 rows through the scalar form of the instruction model (True) instead of the eight-column AVX2 form (default)."""
    lib().orc_set_x3_scalar(C.c_int(1 if on else 0))


def softplus100(x): return _unary('orc_softplus100', x)
def softplus100_lean(x): return _unary('orc_softplus100_lean', x)   # the f32x3 mode's activation (det_math.h::dm_softplus100_lean)
def softplus100_arr(x): return _unary('orc_softplus100_arr', x)      # the branch-free array form the MLP rows use
def expneg(x): return _unary('orc_expneg', x)
def log1p01(x): return _unary('orc_log1p01', x)
def sincos(x): return _unary('orc_sincos', x, 2)
def div_consts(x): return _unary('orc_div', x, 2)
def analytic_sdf(x):
    x = _f(x)
    y = np.empty(x.shape[0], np.float32)
    lib().orc_analytic_sdf(_p(x), C.c_int(x.shape[0]), _p(y))
    return y


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    """OpenMP threads of the C oracle from now on (tools/time_reference_cpu.py times the port at 1 and at all threads)"""
    lib().orc_set_num_threads(C.c_int(int(n)))
