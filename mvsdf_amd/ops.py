"""Thin torch-tensor wrappers over the C ABI (raw pointers + current HIP stream).  Device memory comes from
torch; nothing here computes -- every function is one or a few C calls into libmvsdf_hip.so."""
import ctypes as C

import torch

from . import _lib
from ._lib import NetDesc, TraceParams, check, lib, ptr, stream_of


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda, 'expected a float32 CUDA(HIP) tensor'
    return t.contiguous()


class PackedLayer:
    __slots__ = ('w', 'wp', 'wpT', 'bias', 'K', 'N')


class PackedNet:
    """A weight-norm-folded MLP: row-major W (for autograd bookkeeping), MFMA-packed W and W^T, biases."""

    def __init__(self, layers, skip_layer, multires):
        self.layers, self.skip_layer, self.multires = layers, skip_layer, multires

    def desc(self, transposed=False):
        d = NetDesc()
        d.n_layers = len(self.layers)
        for i, L in enumerate(self.layers):
            d.K[i], d.N[i] = L.K, L.N
            d.wp[i] = (L.wpT if transposed else L.wp).data_ptr()
            d.bias[i] = L.bias.data_ptr()
        d.skip_layer, d.multires = self.skip_layer, self.multires
        return d


def fold_pack(v, g, want_t=True):
    """weight_norm fold + packing: v[N,K], g[N,1] -> (w[N,K], wp, wpT)  (idr.py:70-71)."""
    v, g = _f32(v), _f32(g).reshape(-1)
    N, K = v.shape
    n = lib().mvsdf_packed_floats(N, K)
    w = torch.empty_like(v)
    wp = torch.empty(n, dtype=torch.float32, device=v.device)
    wpT = torch.empty(n, dtype=torch.float32, device=v.device) if want_t else None
    check(lib().mvsdf_fold_pack(ptr(v), ptr(g), N, K, ptr(w), ptr(wp), ptr(wpT), stream_of(v)), 'mvsdf_fold_pack')
    return w, wp, wpT


def fold_backward(v, g, dW):
    v, g, dW = _f32(v), _f32(g).reshape(-1), _f32(dW)
    N, K = v.shape
    dv, dg = torch.empty_like(v), torch.empty_like(g)
    check(lib().mvsdf_fold_backward(ptr(v), ptr(g), ptr(dW), N, K, ptr(dv), ptr(dg), stream_of(v)), 'mvsdf_fold_backward')
    return dv, dg.reshape(-1, 1)


def pack_net(vs, gs, biases, skip_layer, multires, want_t=True):
    layers = []
    for v, g, b in zip(vs, gs, biases):
        L = PackedLayer()
        L.w, L.wp, L.wpT = fold_pack(v, g, want_t)
        L.bias = _f32(b)
        L.N, L.K = v.shape
        layers.append(L)
    return PackedNet(layers, skip_layer, multires)


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


def trace(net, cam_loc, ray_dirs, object_mask, params, training, intervals, minsdf_steps=None, mt=2, rpw=2):
    """RayTracing.forward on the device -> (points[R,3], mask[R] bool, dists[R], counters[16] int64 device tensor)."""
    cam_loc, ray_dirs = _f32(cam_loc), _f32(ray_dirs)
    B, P = ray_dirs.shape[:2]
    R = B * P
    dev = ray_dirs.device
    om = object_mask.reshape(-1).to(torch.uint8).contiguous()
    pts = torch.empty(R, 3, dtype=torch.float32, device=dev)
    mask = torch.empty(R, dtype=torch.uint8, device=dev)
    dists = torch.empty(R, dtype=torch.float32, device=dev)
    counters = torch.empty(16, dtype=torch.int64, device=dev)
    wsb = lib().mvsdf_trace_workspace_bytes(R)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    tp = TraceParams(*params)
    d = net.desc()
    check(lib().mvsdf_trace(C.byref(d), C.byref(tp), ptr(cam_loc), ptr(ray_dirs), ptr(om), B, P, 1 if training else 0,
                            ptr(_f32(intervals)), ptr(_f32(minsdf_steps)) if minsdf_steps is not None else None,
                            ptr(pts), ptr(mask), ptr(dists), ptr(counters), ptr(ws), C.c_size_t(wsb), mt, rpw,
                            stream_of(ray_dirs)), 'mvsdf_trace')
    return pts, mask.bool(), dists, counters


def det_math(op, x):
    x = _f32(x).reshape(-1)
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    check(lib().mvsdf_det_math(op, ptr(x), x.numel(), ptr(y0), ptr(y1), stream_of(x)), 'mvsdf_det_math')
    return y0, y1
