"""How far from an fp64 evaluation (same weights, exact softplus) are: the fp32 k-ascending fmaf chain (the bit-exact engine = the C oracle), and the
bf16-matrix-core engines that carry every activation (and, f32x3, every weight) as bf16 terms?  Run on the GPU box: python3 tools/micro/f32s/acc_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import oracle, oracle_np
from helpers import sdf_packed_net, t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth


def f64_sdf(net32, x):
    class N: pass
    m = N()
    m.W = [w.astype(np.float64) for w in net32.W]; m.b = [b.astype(np.float64) for b in net32.b]
    m.n_layers, m.multires, m.skip_layers = net32.n_layers, net32.multires, net32.skip_in
    return oracle_np.sdf_forward(m, x, need_normal=False)[0][:, 0]


def stats(a, b):
    d = np.abs(a - b)
    return 'max %.3g mean %.3g rms %.3g' % (d.max(), d.mean(), np.sqrt((d ** 2).mean()))


for W in (64, 256, 512):
    sd = synth.make_state_dict(W, 0)
    x = np.random.RandomState(3).uniform(-1.2, 1.2, size=(20000, 3)).astype(np.float32)
    for wmode, tag, modes in ((False, 'fp32 weights', (('f32x3', 3, 3),)), ('weights', 'bf16-rounded weights', (('bf16x2', 2, 1), ('bf16x3', 3, 1)))):
        o = oracle.Net(sd, bf16=wmode)
        ref = f64_sdf(o, x)
        chain = oracle.sdf_forward(o, x, ncols=1)[:, 0].astype(np.float64)
        print('W=%d %s: fp32 fmaf chain (oracle = engine f32) vs fp64: %s' % (W, tag, stats(chain, ref)))
        for name, terms, wt in modes:
            net = ops.pack_bf16_net(sdf_packed_net(sd), terms=terms, weight_terms=wt)
            ys = [ops.sdf_col0(net, t(x), mt=mt).cpu().numpy().astype(np.float64) for mt in (1, 2, 4)]
            same = all(np.array_equal(ys[0], y) for y in ys[1:])
            print('    %-7s vs fp64: %s   | vs chain: %s | row tilings bit-equal: %s' % (name, stats(ys[1], ref), stats(ys[1], chain), same))
