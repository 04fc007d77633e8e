"""Where does k_chain_fwd / k_chain_fwd_x3 (the product default; MVSDF_CHAIN_X3=0 with -DMVSDF_DEV_SWITCHES: the fp32 chain) spend its time?  Side build of the library with -DMV_CHAIN_PROBE (clock stamps between the phases of workgroup 0, per
wave), the bench step on it, the mean per launch.   python tools/chain_probe.py [c2|c3|c5share]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvsdf_amd import build
os.environ['MVSDF_LIB'] = build.build(extra_flags=['-DMV_CHAIN_PROBE'], tag='probe')
import numpy as np
import torch
import bench
from mvsdf_amd._lib import lib
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.model.loss import IDRLoss
from mvsdf_amd.optim import FlatAdam
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict
wl = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != 'wgrad' else 'c2'
dev = torch.device('cuda', 0)
model = IDRNetwork(ConfigDict(synth.model_conf(bench.W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
model = model.to(dev).train()
loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
P_, V_ = bench.WORKLOADS[wl]
inp, gt = bench.make_inputs(dev, 0, 1, P_, V_)
def step():
    opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); opt.backward(lo['loss']); opt.step(grad_cap=2.0)
for _ in range(10): step()
torch.cuda.synchronize()
L = lib()
buf = (C.c_ulonglong * 256)()
L.mv_chain_probe_read(buf, 1)
n = 20
for _ in range(n): step()
torch.cuda.synchronize()
L.mv_chain_probe_read(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(16, 16) * 0.01 / n
names = ['gather + PE + H0', 'V: wait in', 'V: gemm', 'V: wait readers', 'V: epilogue', 'last layer', 'N: wait / start', 'N: prologue (fp32 chain) / side loads (x3)', 'N: wait in', 'N: gemm',
         'N: wait readers', 'N: epilogue', 'normal from g0']
print('workload %s: k_chain_fwd, workgroup 0, us per launch (value chain V, normal chain N)' % wl)
print('%-34s' % 'phase' + ''.join('  w%-4d' % w for w in (0, 1, 4, 5, 8, 12, 15)) + '   max over waves')
for i, nm in enumerate(names):
    print('%-34s' % nm + ''.join(' %6.2f' % a[w, i] for w in (0, 1, 4, 5, 8, 12, 15)) + '   %6.2f' % a[:, i].max())
print('%-34s' % 'sum' + ''.join(' %6.2f' % a[w, :13].sum() for w in (0, 1, 4, 5, 8, 12, 15)))
wn = ['MFMA loop (+ loop back)', 'wait: tiles free', 'stage P, Q (global -> LDS)', 'wait: tiles staged', 'bias sum', 'last MFMA loop', 'slab store']
print('k_wgrad_net, one 64 x 64 block of a 256 x 256 layer (256 rows, two operand pairs = 8 stages), us per launch')
print('%-34s' % 'phase' + ''.join('  w%-4d' % w for w in range(4)))
for i, nm in enumerate(wn):
    print('%-34s' % nm + ''.join(' %6.2f' % a[8 + w, i] for w in range(4)))
print('%-34s' % 'sum' + ''.join(' %6.2f' % a[8 + w, :7].sum() for w in range(4)))

