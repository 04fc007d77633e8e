"""Is the f32x3 engine reproduced BIT FOR BIT by the oracle's model of the bf16 matrix instruction?  python3 tools/micro/f32s/bitexact_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle
from helpers import sdf_packed_net, t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
oracle.lib()
for W, n in ((64, 20000), (256, 6000), (512, 2000)):
    for seed in (0, 5, 9):
        sd = synth.make_state_dict(W, seed)
        x = np.random.RandomState(3 + seed).uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
        x[:8] *= 1e-3; x[8:16] = 0.0
        t0 = time.time()
        ref = oracle.sdf_forward(oracle.Net(sd, bf16='f32x3'), x, ncols=1)[:, 0]
        dt = time.time() - t0
        net = ops.pack_bf16_net(sdf_packed_net(sd), terms=3, weight_terms=3)
        for mt in (1, 2, 4):
            y = ops.sdf_col0(net, t(x), mt=mt).cpu().numpy()
            bad = np.nonzero(y.view(np.uint32) != ref.view(np.uint32))[0]
            print('W=%d seed %d mt=%d: %d of %d outputs differ from the oracle%s  (oracle %.1f s)' % (W, seed, mt, bad.size, n, '' if not bad.size else ' e.g. %d: gpu %.9g oracle %.9g' % (bad[0], y[bad[0]], ref[bad[0]]), dt), flush=True)
