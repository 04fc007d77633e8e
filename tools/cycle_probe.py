"""Dev probe: which objects of a training step are only freed by Python's cyclic collector?  (A forward block that waits for the collector stays allocated for several
steps: 3.6 GB each in the shipped workload.)"""
import gc, os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
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
def step():
    opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); opt.backward(lo['loss']); opt.all_reduce_mean(defer_scale=True); opt.step(grad_cap=2.0, zero_grad=True)
for _ in range(3): step()
torch.cuda.synchronize(); gc.collect(); gc.disable()
a0 = torch.cuda.memory_allocated()
for i in range(6):
    step(); print('step %d: allocated %.1f MB above the start' % (i, (torch.cuda.memory_allocated() - a0) / 1e6))
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
print('collector found %d unreachable objects after 6 steps' % n)
c = collections.Counter(type(o).__name__ for o in gc.garbage)
print(c.most_common(25))
for o in gc.garbage:
    if type(o).__name__ in ('StepRecord', 'PendingOutputs'):
        print(type(o).__name__, 'referrers:', [type(r).__name__ for r in gc.get_referrers(o)][:8])
        break
