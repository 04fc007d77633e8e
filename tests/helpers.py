import numpy as np
import torch

from mvsdf_amd import ops
from mvsdf_amd.utils import synth


def sdf_packed_net(sd, dev='cuda', prefix='implicit_network', skip_layer=4, multires=6):
    vs, gs, bs = [], [], []
    l = 0
    while '%s.lin%d.weight_v' % (prefix, l) in sd:
        vs.append(torch.from_numpy(sd['%s.lin%d.weight_v' % (prefix, l)]).to(dev))
        gs.append(torch.from_numpy(sd['%s.lin%d.weight_g' % (prefix, l)]).to(dev))
        bs.append(torch.from_numpy(sd['%s.lin%d.bias' % (prefix, l)]).to(dev))
        l += 1
    return ops.pack_net(vs, gs, bs, skip_layer, multires)


def trace_params(W=64, **over):
    tr = dict(synth.model_conf(W)['ray_tracer'])
    tr.update(over)
    return (tr['object_bounding_sphere'], tr['sdf_threshold'], tr['line_search_step'], tr['line_step_iters'],
            tr['sphere_tracing_iters'], tr['n_steps'], tr['n_secant_steps'], 0.5)


def t(a, dev='cuda'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
