import sys, os, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import sdf_packed_net, t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
sd = synth.make_state_dict(256, 0)
net = sdf_packed_net(sd)
gen = torch.Generator().manual_seed(5)
for M, r0, nx in ((4500, 100, 2000), (1500, 100, 1000), (1500, 96, 1000)):
    x = (torch.rand(M, 3, generator=gen) * 2 - 1).cuda()
    dy = torch.randn(M, net.layers[-1].N, generator=gen).cuda() * 0.1
    dn = torch.randn(M, 3, generator=gen).cuda()
    y, n, ctx = ops.sdf_forward(net, x, M)
    dWs, dbs, dx = ops.sdf_backward(net, x, M, M, M, dy, dn, ctx, True)
    dWs2, dbs2, dx2 = ops.sdf_backward(net, x, M, M, M, dy, dn, ctx, True)
    wsA, dxX = ops.sdf_backward_pair(net, M, M, M, dy, dn, r0, nx, dy[r0:r0 + nx].contiguous(), dn[r0:r0 + nx].contiguous(), ctx)
    wsA2, dxX2 = ops.sdf_backward_pair(net, M, M, M, dy, dn, r0, nx, dy[r0:r0 + nx].contiguous(), dn[r0:r0 + nx].contiguous(), ctx)
    d = (dxX - dx[r0:r0 + nx]).abs()
    print('M=%d r0=%d: backward repeat equal %s, pair repeat equal %s, X vs full: equal %s max diff %.3g (|dx| max %.3g), rows differing %d' % (
        M, r0, torch.equal(dx, dx2), torch.equal(dxX, dxX2), torch.equal(dxX, dx[r0:r0 + nx]), float(d.max()), float(dx.abs().max()), int((d.max(1).values > 0).sum())))
