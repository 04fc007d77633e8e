"""Host time per ops.* wrapper and per autograd Function of one training step (any thread), bf16 tracer so that the GPU is not the limit."""
import os, sys, time, types, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mvsdf_amd import ops, functional as Fn
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.model.loss import IDRLoss
from mvsdf_amd.optim import FlatAdam
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(mod, name, label):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try: return f(*a, **k)
        finally: acc[label] += time.perf_counter() - t0; cnt[label] += 1
    setattr(mod, name, g)
for n, f in list(vars(ops).items()):
    if isinstance(f, types.FunctionType) and not n.startswith('_') and n not in ('check', 'lib', 'ptr', 'stream_of'): wrap(ops, n, 'ops.' + n)
for cls in (Fn._IdrStep, Fn._FoldNet, Fn._LossTerms, Fn._FeatCorrPP):
    for m in ('forward', 'backward'):
        f = getattr(cls, m)
        def mk(f, label):
            def g(*a, **k):
                t0 = time.perf_counter()
                try: return f(*a, **k)
                finally: acc[label] += time.perf_counter() - t0; cnt[label] += 1
            return staticmethod(g)
        setattr(cls, m, mk(f, 'Fn.%s.%s' % (cls.__name__, m)))
dev = torch.device('cuda', 0)
model = IDRNetwork(ConfigDict(synth.model_conf(bench.W))); model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
model = model.to(dev).train().set_trace_dtype(sys.argv[1] if len(sys.argv) > 1 else 'bf16')
loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
P_, V_ = bench.WORKLOADS['c2']; inp, gt = bench.make_inputs(dev, 0, 1, P_, V_)
def step():
    opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); opt.backward(lo['loss']); opt.all_reduce_mean(defer_scale=True); opt.step(grad_cap=2.0)
for _ in range(10): step()
torch.cuda.synchronize(); acc.clear(); cnt.clear()
n = 50; t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print('host loop %.3f ms/step' % ((t1 - t0) / n * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print('%-36s %6.1f us/step  %4.1f calls/step' % (k, v / n * 1e6, cnt[k] / n))
