import os, sys, time, cProfile, pstats
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
    opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); opt.backward(lo['loss']); opt.all_reduce_mean(defer_scale=True); opt.step(grad_cap=2.0)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr)
# microseconds per step (pstats prints milliseconds with three decimals: too coarse for a 0.4 ms step)
rows = sorted(((tt, ct, nc, '%s:%d(%s)' % (os.path.basename(f), l, fn)) for (f, l, fn), (cc, nc, tt, ct, _) in st.stats.items()), reverse=True)
n = 20
print('%9s %9s %7s  function   (us per step: own time, cumulative; calls per step)' % ('own', 'cum', 'calls'))
for tt, ct, nc, name in rows[:60]:
    print('%9.1f %9.1f %7.1f  %s' % (tt / n * 1e6, ct / n * 1e6, nc / n, name))
