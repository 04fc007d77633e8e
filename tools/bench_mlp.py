"""Perf probe of the tracing-MLP kernel alone (dev tool)."""
import argparse, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import sdf_packed_net, t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
ap = argparse.ArgumentParser()
ap.add_argument('--W', type=int, default=256)
ap.add_argument('--n', default='4096,16384,65536,262144')
ap.add_argument('--mt', default='1,2,4')
ap.add_argument('--bf16', action='store_true')
ap.add_argument('--dtype', default='', help='f32 | f32x3 | bf16 | bf16w | bf16x2 | bf16x3 (overrides --bf16)')
a = ap.parse_args()
net = sdf_packed_net(synth.make_state_dict(a.W, 0), bf16=a.bf16 and not a.dtype)
if a.dtype:
    ops.pack_trace_net(net, a.dtype)
Ft = 2 * sum(i * o for i, o in synth.sdf_layer_dims(a.W)[:-1]) + 2 * synth.sdf_layer_dims(a.W)[-1][0]
for n in [int(v) for v in a.n.split(',')]:
    x = (torch.rand(n, 3, device='cuda') * 2 - 1)
    for mt in [int(v) for v in a.mt.split(',')]:
        for _ in range(3): y = ops.sdf_col0(net, x, mt=mt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = ops.sdf_col0(net, x, mt=mt)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('W=%d n=%d mt=%d: %.3f ms  %.1f TFLOP/s (%.1f%%)  %.1f us per 16-row tile-eval per CU' % (a.W, n, mt, ms, n * Ft / ms / 1e9, n * Ft / ms / 1e9 / 1.573, ms * 1e3 / max(1.0, n / 16 / 256)), flush=True)
