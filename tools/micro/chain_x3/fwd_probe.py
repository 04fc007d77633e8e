"""Where does the forward chain spend its time?  Side build with -DMV_CHAIN_PROBE (clock stamps between the phases of workgroup 0, per wave), the forward
of M rows of the 8x256 SDF network, mean per launch; both arithmetics.   python tools/micro/chain_x3/fwd_probe.py [M]"""
import ctypes as C
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mvsdf_amd import build
XF = os.environ.get('X3_FLAGS', '').split()
os.environ['MVSDF_LIB'] = build.build(extra_flags=['-DMV_CHAIN_PROBE', '-DMVSDF_DEV_SWITCHES'] + XF, tag='probe' + ''.join(c for c in ''.join(XF) if c.isalnum()))
import numpy as np
import torch
from mvsdf_amd import ops
from mvsdf_amd._lib import lib
from mvsdf_amd.utils import synth
from helpers import sdf_packed_net
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3100
sd = synth.make_state_dict(256, 0)
x = (torch.rand(M, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1).cuda()
L = lib()
names = ['gather + PE + H0', 'V: wait in', 'V: gemm', 'V: wait readers', 'V: epilogue', 'last layer', 'N: start', 'N: side loads', 'N: wait in', 'N: gemm',
         'N: wait readers', 'N: epilogue', 'normal from g0']
for flag in (True,) if os.environ.get('X3_ONLY') else (False, True):
    ops.CHAIN_X3 = flag
    net = sdf_packed_net(sd)
    for _ in range(5):
        ops.sdf_forward(net, x, M)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 256)()
    L.mv_chain_probe_read(buf, 1)
    n = 20
    for _ in range(n):
        ops.sdf_forward(net, x, M)
    torch.cuda.synchronize()
    L.mv_chain_probe_read(buf, 0)
    a = np.array(list(buf), dtype=np.float64).reshape(16, 16) * 0.01 / n
    print('%s chain, M = %d: workgroup 0, us per launch (value chain V, normal chain N)' % ('x3' if flag else 'f32', M))
    print('%-34s' % 'phase' + ''.join('  w%-4d' % w for w in (0, 1, 4, 5, 8, 12, 15)) + '   max over waves')
    for i, nm in enumerate(names):
        print('%-34s' % nm + ''.join(' %6.2f' % a[w, i] for w in (0, 1, 4, 5, 8, 12, 15)) + '   %6.2f' % a[:, i].max())
    print('%-34s' % 'sum' + ''.join(' %6.2f' % a[w, :13].sum() for w in (0, 1, 4, 5, 8, 12, 15)))
