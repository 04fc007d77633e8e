"""cProfile of the Python side of the custom backward functions, called directly on the main thread (the autograd engine runs them on
its own thread where cProfile cannot see them)."""
import os, sys, cProfile, pstats, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mvsdf_amd import functional as Fn
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
captured = {}
_orig = Fn._IdrStep.backward
def spy(ctx, *g):
    captured['ctx'], captured['g'] = ctx, g
    return _orig(ctx, *g)
Fn._IdrStep.backward = staticmethod(spy)
for _ in range(3):
    opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); lo['loss'].backward()
torch.cuda.synchronize()
ctx, g = captured['ctx'], captured['g']
for _ in range(5): _orig(ctx, *g)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(50): _orig(ctx, *g)
dt = time.perf_counter() - t0
pr.disable(); torch.cuda.synchronize()
print(f'_IdrStep.backward host time {dt / 50 * 1e3:.3f} ms per call')
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
