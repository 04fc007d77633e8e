"""Step time in phase 0 (train_progress < 1/6: depth-surface sampling active) vs phase 1 on the bench workload."""
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
inp, gt = bench.make_inputs(dev, 0)
print('depths', tuple(inp['depths'].shape))
for tp in (0.3, 0.05):
    def step():
        opt.zero_grad(); out = model(inp, tp); lo = loss_fn(out, dict(gt), tp, bench.B); lo['loss'].backward(); opt.step(grad_cap=2.0 if tp >= 1 / 6 else None)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): step()
    torch.cuda.synchronize()
    print(f'train_progress {tp}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step')
