"""Soak: a few hundred real optimisation steps (lr > 0) on the synthetic batch: finite losses, loss goes down, memory stays flat."""
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
loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=1e-4)
sched = torch.optim.lr_scheduler.MultiStepLR(opt, [200], gamma=0.5)
inp, gt = bench.make_inputs(dev, 0)
n = int(os.environ.get('STEPS', 400))
hist = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(n):
    tp = 0.05 if it < 50 else 0.3                      # a stretch of phase 0 (depth-surface sampling), then phase 1
    opt.zero_grad(); out = model(inp, tp); lo = loss_fn(out, dict(gt), tp, bench.B); lo['loss'].backward(); opt.step(grad_cap=2.0 if tp >= 1 / 6 else None)
    sched.step()
    if it % 50 == 0 or it == n - 1:
        hist.append((it, float(lo['loss']), float(lo['rgb_loss']), float(lo['eikonal_loss']), int(out['network_object_mask'].sum()),
                     float(opt.grad_norm()), torch.cuda.memory_allocated() / 2**20, torch.cuda.max_memory_allocated() / 2**20))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
for h in hist: print('step %4d  loss %.4f  rgb %.4f  eik %.4f  hits %4d  |g| %.3f  mem %.0f MiB (peak %.0f)' % h)
print(f'{n} steps in {dt:.2f} s = {dt / n * 1e3:.2f} ms/step; all finite: {all(map(lambda h: h[1] == h[1] and abs(h[1]) < 1e6, hist))}')
