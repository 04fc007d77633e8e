"""ctypes binding of libmvsdf_hip.so (C ABI: include/mvsdf_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, we raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get('MVSDF_LIB') or os.path.join(_HERE, 'libmvsdf_hip.so')   # MVSDF_LIB: dev override (ablation builds)
MAX_LAYERS = 12
_lib = None


class NetDesc(C.Structure):
    _fields_ = [('n_layers', C.c_int), ('K', C.c_int * MAX_LAYERS), ('N', C.c_int * MAX_LAYERS),
                ('wp', C.c_void_p * MAX_LAYERS), ('bias', C.c_void_p * MAX_LAYERS), ('w', C.c_void_p * MAX_LAYERS),
                ('skip_layer', C.c_int), ('multires', C.c_int), ('wp16', C.c_void_p * MAX_LAYERS), ('trace_dtype', C.c_int), ('skip_mask', C.c_uint), ('wx3', C.c_void_p * MAX_LAYERS)]


class TraceParams(C.Structure):
    _fields_ = [('r', C.c_float), ('thr', C.c_float), ('line_search_step', C.c_float), ('line_step_iters', C.c_int),
                ('st_iters', C.c_int), ('n_steps', C.c_int), ('n_secant', C.c_int), ('dist_clip', C.c_float)]


class MvsdfError(RuntimeError):
    pass


def lib():
    """Load the HIP library (built in-tree by mvsdf_amd/build.py).  Raises if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise MvsdfError('libmvsdf_hip.so not found at %s -- run `python -c "import __graft_entry__ as g; g.build()"` '
                             '(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback for the hot path' % SO_PATH)
        import torch  # noqa: F401  -- load torch's bundled HIP runtime first so this library binds to the same libamdhip64
        L = C.CDLL(SO_PATH)
        L.mvsdf_last_error.restype = C.c_char_p
        L.mvsdf_packed_floats.restype = C.c_size_t
        L.mvsdf_packed_floats.argtypes = [C.c_int, C.c_int]
        L.mvsdf_trace_workspace_bytes.restype = C.c_size_t
        L.mvsdf_trace_workspace_bytes.argtypes = [C.c_int]
        L.mvsdf_trace_workspace_bytes_n.restype = C.c_size_t
        L.mvsdf_trace_workspace_bytes_n.argtypes = [C.c_int, C.c_int]
        for fn in ('mvsdf_sdf_ctx_floats', 'mvsdf_sdf_bwd_ws_floats', 'mvsdf_render_ctx_floats', 'mvsdf_render_bwd_ws_floats'):
            getattr(L, fn).restype = C.c_size_t
        L.mvsdf_adam_ws_floats.restype = C.c_size_t
        L.mvsdf_packed_bf16_bytes.restype = C.c_size_t
        L.mvsdf_packed_bf16_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        L.mvsdf_tracegen_state_bytes.restype = C.c_size_t
        L.mvsdf_adam_step.argtypes = [C.c_void_p] * 4 + [C.c_size_t] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mvsdf_adam_step_scaled.argtypes = [C.c_void_p] * 4 + [C.c_size_t] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mvsdf_adam_step_fused.argtypes = [C.c_void_p] * 4 + [C.c_size_t] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mvsdf_loss_scale.argtypes = [C.c_void_p] + [C.c_float] * 5 + [C.c_void_p, C.c_void_p, C.c_int] * 4 + [C.c_void_p, C.c_void_p]   # g: host array of 6 pointers
        for name in EXPORTS:
            getattr(L, name)
        _lib = L
    return _lib


# every symbol include/mvsdf_hip.h declares (tests/test_abi.py checks the header against this list and the .so)
EXPORTS = [
    'mvsdf_version', 'mvsdf_abi_struct_sizes', 'mvsdf_last_error', 'mvsdf_packed_floats', 'mvsdf_fold_pack', 'mvsdf_fold_backward', 'mvsdf_fold_pack_net', 'mvsdf_fold_backward_net', 'mvsdf_packed_bf16_bytes', 'mvsdf_pack_bf16w_net', 'mvsdf_pack_bf16s_net', 'mvsdf_pack_bf16x3_net', 'mvsdf_pack_bf16x3t_net',
    'mvsdf_sdf_col0', 'mvsdf_camera_rays', 'mvsdf_sphere_intersection', 'mvsdf_trace_workspace_bytes', 'mvsdf_trace_workspace_bytes_n', 'mvsdf_trace', 'mvsdf_trace_stage', 'mvsdf_det_math',
    'mvsdf_tracegen_state_bytes', 'mvsdf_tracegen_init', 'mvsdf_tracegen_step', 'mvsdf_tracegen_finish', 'mvsdf_tracegen_rows', 'mvsdf_tracegen_reduce',
    'mvsdf_tracegen_secant',
    'mvsdf_sdf_ctx_floats', 'mvsdf_sdf_forward', 'mvsdf_sdf_bwd_ws_floats', 'mvsdf_sdf_backward', 'mvsdf_sdf_backward_pair', 'mvsdf_sdf_backward_finish',
    'mvsdf_feat_corr', 'mvsdf_depth_carve', 'mvsdf_loss_terms', 'mvsdf_loss_prep', 'mvsdf_loss_scale', 'mvsdf_adam_ws_floats', 'mvsdf_adam_step', 'mvsdf_adam_step_scaled', 'mvsdf_adam_step_fused',
    'mvsdf_partition_rays', 'mvsdf_step_outputs', 'mvsdf_step_backward_inputs', 'mvsdf_step_backward_fbar', 'mvsdf_dsurf_select', 'mvsdf_dsurf_points',
    'mvsdf_render_ctx_floats', 'mvsdf_render_bwd_ws_floats', 'mvsdf_render_forward', 'mvsdf_render_backward',
    'mvsdf_step_create', 'mvsdf_step_destroy', 'mvsdf_step_forward', 'mvsdf_step_wait_counts', 'mvsdf_step_backward', 'mvsdf_step_set_timing', 'mvsdf_step_trace_times', 'mvsdf_step_times',
    'mvsdf_step_seq', 'mvsdf_step_counts_offset', 'mvsdf_step_wait_counts_seq', 'mvsdf_step_done_seq', 'mvsdf_step_can_defer', 'mvsdf_step_saved_offsets',
    'mvsdf_loss_layout', 'mvsdf_loss_forward', 'mvsdf_loss_backward',
]


def check(rc, what=''):
    if rc != 0:
        raise MvsdfError('%s failed (code %d): %s' % (what or 'mvsdf call', rc, lib().mvsdf_last_error().decode()))


def ptr(t):
    """device (or host) pointer of a contiguous torch tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor must be contiguous'
    return C.c_void_p(t.data_ptr())


def stream_of(t):
    import torch
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)
