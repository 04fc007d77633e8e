"""Wall-clock split of one training step (dev tool): forward / loss / backward / optimiser, with and without syncs."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.model.loss import IDRLoss
from mvsdf_amd.parallel import FlatGradBucket
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict
dev = torch.device('cuda', 0)
model = IDRNetwork(ConfigDict(synth.model_conf(bench.W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
model = model.to(dev).train()
loss_fn = IDRLoss(); bucket = FlatGradBucket(model.parameters()); opt = torch.optim.Adam(model.parameters(), lr=0.0)
inp, gt = bench.make_inputs(dev, 0)
def sync(): torch.cuda.synchronize()
for it in range(8):
    sync(); t0 = time.perf_counter()
    bucket.zero(); out = model(inp, bench.TP)
    t1h = time.perf_counter(); sync(); t1 = time.perf_counter()
    lo = loss_fn(out, dict(gt), bench.TP, bench.B)
    t2h = time.perf_counter(); sync(); t2 = time.perf_counter()
    lo['loss'].backward()
    t3h = time.perf_counter(); sync(); t3 = time.perf_counter()
    bucket.clip_(2.0); opt.step()
    t4h = time.perf_counter(); sync(); t4 = time.perf_counter()
    if it >= 3:
        print('fwd %.2f (host %.2f)  loss %.2f (host %.2f)  bwd %.2f (host %.2f)  opt %.2f (host %.2f)  total %.2f ms' % (
            (t1-t0)*1e3, (t1h-t0)*1e3, (t2-t1)*1e3, (t2h-t1)*1e3, (t3-t2)*1e3, (t3h-t2)*1e3, (t4-t3)*1e3, (t4h-t3)*1e3, (t4-t0)*1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for it in range(5):
    bucket.zero(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); lo['loss'].backward(); bucket.clip_(2.0); opt.step()
sync(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
