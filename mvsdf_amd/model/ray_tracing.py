"""RayTracing with the reference's constructor and forward signature (reference code/model/ray_tracing.py:7-98), executed by the HIP tracer
(csrc/trace.hip).

Two routes, same per-ray state machine:
  * `sdf` carries the folded network (`ImplicitNetwork.native_sdf()`, what IDRNetwork passes): the fused path -- 7 launches per call
    (k_sphere_trace, 3 x k_ray_samples, 3 x k_reduce_items), MLP evaluated inside the kernels, no host synchronisation;
  * `sdf` is any other callable [M,3] -> [M] (the reference's signature, ray_tracing.py:27-32): the generic path -- the kernels emit the
    points to evaluate, the callable runs on the host side of the API between launches (one round trip per evaluation), the kernels
    consume its values.  Slow by construction, bit-identical decisions."""
import os

import torch
import torch.nn as nn

from .. import ops
from ..utils.general import PinnedUniform


class NativeSDF:
    """The `sdf` callable handed to RayTracing.forward by IDRNetwork: callable like the reference's lambda
    (idr.py:194) but also carries the folded, MFMA-packed weights the native tracer needs."""

    def __init__(self, packed_net):
        self.native_net = packed_net

    def __call__(self, x):
        return ops.sdf_col0(self.native_net, x)


class RayTracing(nn.Module):
    def __init__(self, object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=1,
                 sphere_tracing_iters=10, n_steps=100, n_secant_steps=8):
        super().__init__()
        self.object_bounding_sphere = object_bounding_sphere
        self.sdf_threshold = sdf_threshold
        self.sphere_tracing_iters = sphere_tracing_iters
        self.line_step_iters = line_step_iters
        self.line_search_step = line_search_step
        self.n_steps = n_steps
        self.n_secant_steps = n_secant_steps
        self.last_counters = None           # device int64[16]: MLP rows per stage (include/mvsdf_hip.h MVSDF_CNT_*)
        self.mt = None                      # row tiles per sphere-tracing workgroup (None: pick from the ray count)
        self.mt_samples = None              # row tiles per chunk of the sample-row kernels
        self._intervals = None
        self._draw = PinnedUniform()
        self.events = None                  # set to a list to have per-kernel (start, mid, end) events appended each call

    def _params(self):
        if os.environ.get('IDR_USE_ENV', '0') == '1' and os.environ.get('IDR_RENDER', '0') == '1':
            dist_clip, iters = 0.05, 40                                        # ray_tracing.py:127-131
        else:
            dist_clip, iters = 0.5, self.sphere_tracing_iters
        return (self.object_bounding_sphere, self.sdf_threshold, self.line_search_step, self.line_step_iters, iters,
                self.n_steps, self.n_secant_steps, dist_clip)

    def intervals(self, dev):
        """torch.linspace(0, 1, n_steps) on the device (ray_tracing.py:206; CPU values like the reference), uploaded once per (n_steps, device)."""
        key = (self.n_steps, str(dev))
        if self._intervals is None or self._intervals[0] != key:
            self._intervals = (key, torch.linspace(0, 1, steps=self.n_steps).to(dev))
        return self._intervals[1]

    def tiling(self, R, trace_dtype=0):
        """(row tiles per sphere-tracing workgroup, row tiles per chunk of the sample-row kernels) for R rays.  Rays per sphere-tracing
        workgroup = 8 * mt: one workgroup per CU (256 of them) while the batch allows it -- the kernel is a chain of dependent evaluations, so
        fewer, fuller workgroups beat two contending ones per CU (4096 rays: 4.17 -> 3.97 ms per step with mt = 2); 4 tiles once the chip is
        over-subscribed anyway (finer compaction of the rays still active; +6 % at 8k-32k rays)."""
        mt = self.mt or (1 if R <= 2048 else (2 if R <= 4096 else 4))                                 # (self.mt / self.mt_samples: overrides for sweeps)
        # sample-row kernels: two row tiles per workgroup, two workgroups per CU; the three-weight-term engine ('f32x3', trace_dtype 5) streams 1.5x the
        # fp32 pack per evaluation: at 2048 rays four tiles, one workgroup per CU (c2 1.60 -> 1.53 ms; 4096 rays 2.27 -> 2.31, 8192 rays equal)
        mt_samples = self.mt_samples or (4 if (trace_dtype == 5 and R <= 2048) else 2)
        return mt, mt_samples

    def forward(self, sdf, cam_loc, object_mask, ray_directions, minsdf_steps=None, mask_ready=None, defer_minsdf=None):
        """-> (points[R,3], network_object_mask[R] bool, dists[R]).
        minsdf_steps: the n_steps uniform draws of minimal_sdf_points (ray_tracing.py:287); drawn here from torch's CPU
        generator when not given -- always, whereas the reference draws only if some ray needs them.
        mask_ready: optional callable(network_object_mask) run once the mask is final, before the secant / min-sdf launch is enqueued
        (IDRNetwork uses it to fetch the hit count while that launch runs).
        defer_minsdf: see ops.trace (IDRNetwork.lazy_unused_outputs)."""
        net = getattr(sdf, 'native_net', None)
        dev = ray_directions.device
        intervals = self.intervals(dev)
        if self.training and minsdf_steps is None:
            minsdf_steps = self._draw((self.n_steps,), 0.0, 1.0, dev)           # CPU generator, pinned staging, async copy
        elif minsdf_steps is not None:
            minsdf_steps = minsdf_steps.to(dev, non_blocking=True)
        if net is None:                                                       # opaque callable: emit / evaluate / consume rounds
            with torch.no_grad():
                pts, mask, dists, counters = ops.trace_generic(sdf, cam_loc, ray_directions, object_mask, self._params(), self.training,
                                                               intervals, minsdf_steps)
            self.last_counters = counters
            if mask_ready is not None:
                mask_ready(mask)
            return pts, mask, dists
        R = ray_directions.shape[0] * ray_directions.shape[1]
        mt, mt_samples = self.tiling(R, getattr(net, 'trace_dtype', 0))
        pts, mask, dists, counters = ops.trace(net, cam_loc, ray_directions, object_mask, self._params(), self.training, intervals,
                                               minsdf_steps, mt=mt, mt_samples=mt_samples, events=self.events,
                                               mask_ready=mask_ready, defer_minsdf=defer_minsdf)
        self.last_counters = counters
        return pts, mask, dists
