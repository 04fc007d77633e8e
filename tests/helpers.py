import numpy as np
import torch

from mvsdf_amd import ops
from mvsdf_amd.utils import synth


def sdf_packed_net(sd, dev='cuda', prefix='implicit_network', skip_layer=4, multires=6, bf16=False):
    vs, gs, bs = [], [], []
    l = 0
    while '%s.lin%d.weight_v' % (prefix, l) in sd:
        vs.append(torch.from_numpy(sd['%s.lin%d.weight_v' % (prefix, l)]).to(dev))
        gs.append(torch.from_numpy(sd['%s.lin%d.weight_g' % (prefix, l)]).to(dev))
        bs.append(torch.from_numpy(sd['%s.lin%d.bias' % (prefix, l)]).to(dev))
        l += 1
    net = ops.pack_net(vs, gs, bs, skip_layer, multires)
    return ops.pack_bf16_net(net) if bf16 else net


def trace_params(W=64, **over):
    tr = dict(synth.model_conf(W)['ray_tracer'])
    tr.update(over)
    return (tr['object_bounding_sphere'], tr['sdf_threshold'], tr['line_search_step'], tr['line_step_iters'],
            tr['sphere_tracing_iters'], tr['n_steps'], tr['n_secant_steps'], 0.5)


def t(a, dev='cuda'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def analytic_sdf(x):
    """The tier-0 SDF of tests/golden/make_golden.py::analytic_sdf (built from +, -, * only: every torch elementwise op is one correctly
    rounded IEEE operation on CPU and GPU alike, so the GPU evaluation is bit-identical to the one the goldens were made with)."""
    X, Y, Z = x[:, 0], x[:, 1], x[:, 2]
    x2, y2, z2 = X * X, Y * Y, Z * Z
    r2 = x2 + y2 + z2
    t5 = ((16.0 * x2 - 20.0) * x2 + 5.0) * X
    t4 = (8.0 * y2 - 8.0) * y2 + 1.0
    t3 = (4.0 * z2 - 3.0) * Z
    return 1.4 * (r2 - 0.36) + 0.2 * (t5 * t4 * t3)
