"""idr_w512 fixture batch, Python route: rendering lin0 bias gradient with the x3 chains vs the fp32 chains -- sparse differences = ReLU mask flips."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import golden
from helpers import t
from mvsdf_amd import ops
from mvsdf_amd.model.loss import IDRLoss
from mvsdf_amd.utils import synth
from test_gpu_idr import build
g = golden('idr_w512')
W, B, P, V, seed, tp = int(g['W']), int(g['B']), int(g['P']), int(g['V']), int(g['seed']), float(g['tp'])
res = {}
for flag in (True, False):
    ops.CHAIN_X3 = flag
    model, sd = build(W, seed)
    model.native_step = False
    inp, gt = synth.make_batch(B, P, V, seed=seed, size=float(g['scene_size']), center=tuple(g['scene_center']), feat_hw=tuple(int(v) for v in g['feat_hw']), focal_scale=float(g['focal_scale']))
    model.train(); torch.manual_seed(seed + 5)
    out = model({k: t(v) for k, v in inp.items()}, tp)
    lo = IDRLoss()(out, {k: t(v) for k, v in gt.items()}, tp, B)
    model.zero_grad(); lo['loss'].backward()
    res[flag] = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()}
for k in ('rendering_network.lin0.bias', 'rendering_network.lin1.bias', 'implicit_network.lin8.bias'):
    a, b = res[True][k], res[False][k]
    d = np.abs(a - b)
    order = np.argsort(-d)[:6]
    print(k, 'max |g| %.3g; largest differences:' % np.abs(b).max(), ' '.join('%d:%.2e' % (i, d[i]) for i in order), '| median difference %.2e' % np.median(d))
