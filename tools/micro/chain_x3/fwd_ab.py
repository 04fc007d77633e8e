"""A/B of the fused forward chain (value + normal) of the SDF network: fp32-input MFMA (k_chain_fwd) vs the three-term bf16 form (k_chain_fwd_x3), same
weights, same rows: max deviation of every saved tensor against a float64 evaluation (oracle/oracle_np.py: test infrastructure) and launch time."""
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


def run(W, M, skips=(4,), reps=30):
    sd = synth.make_state_dict(W, 0, skip_in=skips)
    x = (torch.rand(M, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1).cuda()
    res = {}
    for name, flag in (('f32', False), ('x3', True)):
        ops.CHAIN_X3 = flag
        net = sdf_packed_net(sd, skip_layer=skips if len(skips) != 1 else skips[0])
        y, n, ctx = ops.sdf_forward(net, x, M)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.sdf_forward(net, x, M)
        torch.cuda.synchronize()
        res[name] = (y.cpu().numpy(), n.cpu().numpy(), (time.perf_counter() - t0) / reps * 1e6)
    onet = ON.sdf_net(sd, skip_in=skips)
    y64, n64, _ = ON.sdf_forward(onet, x.cpu().numpy().astype(np.float64))
    for name in res:
        y, n, us = res[name]
        print('W=%d M=%d skips=%s %-4s: %.1f us | y vs f64 max %.3g (|y| max %.3g) | n vs f64 max %.3g' % (
            W, M, skips, name, us, np.abs(y - y64).max(), np.abs(y64).max(), np.abs(n - n64).max()))
    print('   x3 vs f32: y max %.3g, n max %.3g' % (np.abs(res['x3'][0] - res['f32'][0]).max(), np.abs(res['x3'][1] - res['f32'][1]).max()))


if __name__ == '__main__' and len(sys.argv) > 2:
    run(int(sys.argv[1]), int(sys.argv[2]), reps=10)
elif __name__ == '__main__':
    run(64, 700)
    run(64, 300, (3, 6))
    run(64, 300, (8,))
    run(256, 3100)
    run(256, 6200)
    run(256, 12400)
    run(512, 3100)
