"""A few steps at width W (env W) for rocprofv3."""
import os, sys
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
W = int(os.environ.get('W', 512))
model = IDRNetwork(ConfigDict(synth.model_conf(W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0).items()})
model = model.to(dev).train()
loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
inp, gt = bench.make_inputs(dev, 0)
for _ in range(20):
    opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); lo['loss'].backward(); opt.step(grad_cap=2.0)
torch.cuda.synchronize()
