"""Whole-step gradients, Python route: x3 chains vs fp32 chains, per parameter."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from mvsdf_amd import ops
from test_gpu_native_step import _run
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kw = dict(W=W, B=8, P=128, V=2, tp=0.3, sink=False)
ops.CHAIN_X3 = True
o_n, l_n, g_n, _, m = _run(False, **kw)
ops.CHAIN_X3 = False
o_p, l_p, g_p, _, _ = _run(False, **kw)
for k in o_p:
    d = (o_n[k].float() - o_p[k].float()).abs().max().item() if o_p[k].numel() else 0
    print('out %-28s max |d| %.3g' % (k, d))
off = 0
for name, p in m.named_parameters():
    n = p.numel()
    a, b = g_n[off:off + n], g_p[off:off + n]
    print('%-40s |g| %.3g  max |d| %.3g' % (name, b.abs().max().item(), (a - b).abs().max().item()))
    off += n
