"""Host-side time per section of one training step (no extra syncs): where the CPU thread spends its time."""
import os, sys, time
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
if len(sys.argv) > 1 and sys.argv[1] == 'bf16': model.set_trace_dtype('bf16')
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
_item = torch.Tensor.item
def timed_item(self):
    t0 = time.perf_counter(); r = _item(self); acc['  (item wait)'] = acc.get('  (item wait)', 0.0) + time.perf_counter() - t0; return r
_evs = torch.cuda.Event.synchronize
def timed_evs(self):
    t0 = time.perf_counter(); r = _evs(self); acc['  (count-event wait)'] = acc.get('  (count-event wait)', 0.0) + time.perf_counter() - t0; return r
from mvsdf_amd import native_step as NS
_wc = NS.NativeStep.wait_counts
def timed_wc(self):
    t0 = time.perf_counter(); r = _wc(self); acc['  (count-event wait)'] = acc.get('  (count-event wait)', 0.0) + time.perf_counter() - t0; return r
def step():
    t = time.perf_counter()
    opt.zero_grad(); t = tick('zero', t)
    out = model(inp, bench.TP); t = tick('forward (incl. item wait)', t)
    lo = loss_fn(out, dict(gt), bench.TP, bench.B); t = tick('loss', t)
    opt.backward(lo['loss']); t = tick('backward', t)
    opt.all_reduce_mean(defer_scale=True); opt.step(grad_cap=2.0); t = tick('allreduce+clip+adam', t)
for _ in range(10): step()
torch.cuda.synchronize(); acc.clear()
torch.Tensor.item = timed_item
torch.cuda.Event.synchronize = timed_evs
NS.NativeStep.wait_counts = timed_wc
n = 50
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
for k, v in acc.items(): print(f'{k:32s} {v / n * 1e3:7.3f} ms')
wait = sum(v for k, v in acc.items() if k.startswith('  ('))
print(f'host loop {(t1 - t0) / n * 1e3:.3f} ms/step, of which waiting for the GPU {wait / n * 1e3:.3f} ms -> host work {((t1 - t0) - wait) / n * 1e3:.3f} ms/step; final drain {(t2 - t1) * 1e3:.3f} ms')
