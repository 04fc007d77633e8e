"""A/B of the fused backward pass of the SDF network (E.1 + E.2 + input adjoint, then the weight gradients): fp32-input MFMA chain vs the three-term bf16 chain,
same forward context arithmetic per side; deviations against the float64 oracle (oracle/oracle_np.py: test infrastructure)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mvsdf_amd import ops  # noqa: E402
from mvsdf_amd.utils import synth  # noqa: E402
from helpers import sdf_packed_net  # noqa: E402
from oracle import oracle_np as ON  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def run(W, M, skips=(4,), reps=20):
    sd = synth.make_state_dict(W, 0, skip_in=skips)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(M, 3, generator=g) * 2 - 1).cuda()
    dy = (torch.randn(M, 258, generator=g) * 0.1).cuda()
    dn = torch.randn(M, 3, generator=g).cuda()
    onet = ON.sdf_net(sd, skip_in=skips)
    y64, n64, cache = ON.sdf_forward(onet, x.cpu().numpy().astype(np.float64))
    dW64, db64, dx64 = ON.sdf_backward(onet, cache, dy.cpu().numpy(), dn.cpu().numpy())
    for name, flag in (('f32', False), ('x3', True)):
        ops.CHAIN_X3 = flag
        net = sdf_packed_net(sd, skip_layer=skips if len(skips) != 1 else skips[0])
        y, n, ctx = ops.sdf_forward(net, x, M)
        dWs, dbs, dx = ops.sdf_backward(net, x, M, M, M, dy, dn, ctx, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.sdf_backward(net, x, M, M, M, dy, dn, ctx, True)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        print('W=%d M=%d skips=%s %-4s: %.1f us | dx rel %.3g | dW rel per layer %s | db rel max %.3g' % (
            W, M, skips, name, us, rel(dx.cpu().numpy(), dx64), ' '.join('%.1e' % rel(a.cpu().numpy(), b) for a, b in zip(dWs, dW64)),
            max(rel(a.cpu().numpy(), b) for a, b in zip(dbs, db64))))


if __name__ == '__main__':
    run(64, 700)
    run(64, 300, (8,))
    run(256, 3100)
    run(256, 6200)
    run(512, 1500)
    run(64, 301)            # odd row counts x odd widths: the saved tensors' bases are only 4-byte aligned
    run(64, 77, (3, 6))
