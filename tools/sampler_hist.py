"""Where does the first sign change fall among the 100 ray-sampler samples?  (reads the tracer workspace of one forward)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mvsdf_amd import ops
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict
dev = torch.device('cuda', 0)
keep = {}
_empty = torch.empty
def spy(*a, **k):
    t = _empty(*a, **k)
    if k.get('dtype') == torch.uint8 and t.numel() > 100000: keep['ws'] = t
    return t
W = int(os.environ.get('W', bench.W))
model = IDRNetwork(ConfigDict(synth.model_conf(W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, int(os.environ.get('SEED', 0))).items()})
model = model.to(dev).train()
inp, gt = bench.make_inputs(dev, 0)
net = model.implicit_network.fold()[0]
rd, cl = model.ray_tracer, None
from mvsdf_amd.utils import rend_util
ray_dirs, cam_loc = rend_util.get_camera_params(inp['uv'], inp['pose'], inp['intrinsics'])
R = ray_dirs.shape[0] * ray_dirs.shape[1]
om = torch.ones(R, dtype=torch.bool, device=dev)
torch.empty = spy
# stage 1 + 3 only: sv then holds the sampler values
params = rd._params()
intervals = torch.linspace(0, 1, steps=rd.n_steps).to(dev)
steps = torch.rand(rd.n_steps).to(dev)
def stop(mask): raise StopIteration
try:
    ops.trace(net, cam_loc, ray_dirs, om, params, True, intervals, steps, mt=1, mt_samples=2, mask_ready=stop)
except StopIteration:
    pass
torch.empty = _empty
torch.cuda.synchronize()
ws = keep['ws'].view(torch.float32)
n = rd.n_steps
sv = ws[9 * R: 9 * R + R * n].view(R, n).cpu().numpy()
print('note: counters unavailable here; scanning rows until sv is garbage')
first = []
for k in range(R):
    row = sv[k]
    if not np.all(np.isfinite(row)) or np.all(row == 0): break
    neg = np.nonzero(row < 0)[0]
    first.append(int(neg[0]) if len(neg) else -1)
first = np.array(first)
print('listed rays', len(first), 'with sign change', int((first >= 0).sum()))
f = first[first >= 0]
print('percentiles of first negative index: ', {p: int(np.percentile(f, p)) for p in (10, 25, 50, 75, 90, 95, 99)})
for cut in (4, 8, 12, 16, 24, 32, 48, 64):
    print(f'resolved within first {cut:3d} samples: {(f < cut).mean():.3f}')
