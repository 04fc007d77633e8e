"""Which framework ops of a training step are memory copies (torch.profiler, one step)?"""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
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
def step():
    opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); lo['loss'].backward(); opt.step(grad_cap=2.0)
for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.events():
    n = e.name
    if any(k in n for k in ('copy_', 'Memcpy', 'aten::to', 'aten::contiguous', 'aten::clone', 'aten::fill_', 'aten::zero_', 'aten::index', 'aten::cat', 'aten::mul', 'aten::ones', 'aten::full')) and e.device_type.name == 'CPU':
        st = [s for s in (e.stack or []) if 'mvsdf_amd' in s or 'bench' in s or 'tools' in s][:2]
        print(f'{n:28s} {str(e.input_shapes)[:40]:40s}', ' <- '.join(s.split('/')[-1] for s in st))
