"""Dev probe: host time of IDRNetwork.forward in the shipped workload (8 views x 4096 px, 8x512), step by step with a device sync after every step, then a
cProfile of six such forwards.  (Two profile runs of the round read 13.9 ms per forward, a direct run 0.3-0.5 ms.)"""
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
W = 512
model = IDRNetwork(ConfigDict(synth.model_conf(W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0).items()})
model = model.to(dev).train()
model.set_trace_dtype('f32x3')
loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
P_, V_ = bench.WORKLOADS['shipped']
inp, gt = bench.make_inputs(dev, 0, 1, P_, V_)
def step(t=None):
    opt.zero_grad()
    t0 = time.perf_counter(); out = model(inp, bench.TP); t1 = time.perf_counter()
    lo = loss_fn(out, dict(gt), bench.TP, bench.B); opt.backward(lo['loss']); opt.all_reduce_mean(defer_scale=True); opt.step(grad_cap=2.0, zero_grad=True)
    if t is not None: t.append((t1 - t0) * 1e3)
    if mem is not None: mem.append((model._last_step.layout.fwd_bytes / 1e9, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
mem = None
t = []; mem = []
for _ in range(12): step(t)
torch.cuda.synchronize()
print('forward block GB | allocated GB | reserved GB after each of those steps:', ' '.join('%.2f|%.1f|%.1f' % m for m in mem)); mem = None
print('forward host ms, 12 steps back to back (host runs ahead):', ' '.join('%.2f' % v for v in t))
t = []
for _ in range(12): step(t); torch.cuda.synchronize()
print('forward host ms, 12 steps with a sync after each       :', ' '.join('%.2f' % v for v in t))
pr = cProfile.Profile()
for _ in range(6):
    opt.zero_grad(); pr.enable(); out = model(inp, bench.TP); pr.disable()
    lo = loss_fn(out, dict(gt), bench.TP, bench.B); opt.backward(lo['loss']); opt.all_reduce_mean(defer_scale=True); opt.step(grad_cap=2.0, zero_grad=True); torch.cuda.synchronize()
st = pstats.Stats(pr)
rows = sorted(((tt, ct, nc, '%s:%d(%s)' % (os.path.basename(f), l, fn)) for (f, l, fn), (cc, nc, tt, ct, _) in st.stats.items()), reverse=True)
print('%9s %9s %7s  function   (us per forward: own time, cumulative; calls per forward)' % ('own', 'cum', 'calls'))
for tt, ct, nc, name in rows[:14]:
    print('%9.1f %9.1f %7.1f  %s' % (tt / 6 * 1e6, ct / 6 * 1e6, nc / 6, name))
