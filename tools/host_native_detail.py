"""Host time of a native step in detail: the C calls themselves (ctypes call -> return), the count wait, and the Python around them.
python tools/host_native_detail.py [bf16]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mvsdf_amd import native_step as NS
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.model.loss import IDRLoss
from mvsdf_amd.optim import FlatAdam
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict
dev = torch.device('cuda', 0)
model = IDRNetwork(ConfigDict(synth.model_conf(bench.W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
model = model.to(dev).train()
loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
P_, V_ = bench.WORKLOADS['c2']
inp, gt = bench.make_inputs(dev, 0, 1, P_, V_)
if len(sys.argv) > 1 and sys.argv[1] == 'bf16': model.set_trace_dtype('bf16')
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, g)
L = NS._bind()
class LibProxy:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, n):
        f = getattr(self._lib, n)
        if not n.startswith('mvsdf_'): return f
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); acc['C ' + n] = acc.get('C ' + n, 0.0) + time.perf_counter() - t0; return r
        return g
proxy = LibProxy(L)
NS.lib = lambda: proxy
NS._bind = lambda: proxy
import mvsdf_amd.optim as OPT
OPT.lib = lambda: proxy
wrap(NS._NativeStepFn, 'forward', 'py _NativeStepFn.forward (incl. C + wait)')
wrap(NS._NativeStepFn, 'backward', 'py _NativeStepFn.backward (incl. C)')
wrap(NS._NativeLossFn, 'forward', 'py _NativeLossFn.forward (incl. C)')
wrap(NS._NativeLossFn, 'backward', 'py _NativeLossFn.backward (incl. C)')
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
def step():
    t = time.perf_counter()
    opt.zero_grad(); t = tick('S zero_grad', t)
    out = model(inp, bench.TP); t = tick('S forward', t)
    lo = loss_fn(out, dict(gt), bench.TP, bench.B); t = tick('S loss', t)
    opt.backward(lo['loss']); t = tick('S backward', t)
    opt.all_reduce_mean(defer_scale=True); opt.step(grad_cap=2.0); t = tick('S adam', t)
for _ in range(10): step()
torch.cuda.synchronize(); acc.clear()
n = 100
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
for k in sorted(acc): print(f'{k:52s} {acc[k] / n * 1e3:7.3f} ms')
w = acc.get('C mvsdf_step_wait_counts', 0.0)
print(f'host loop {(t1 - t0) / n * 1e3:.3f} ms/step; waiting {w / n * 1e3:.3f}; host work {((t1 - t0) - w) / n * 1e3:.3f} ms/step')
