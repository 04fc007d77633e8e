"""Sphere-tracing kernel time vs line_step_iters (how much of the dependent chain are deeper line-search back-offs?)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.model.ray_tracing import NativeSDF
from mvsdf_amd.utils import synth, rend_util
from mvsdf_amd.utils.config import ConfigDict
dev = torch.device('cuda', 0)
model = IDRNetwork(ConfigDict(synth.model_conf(bench.W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
model = model.to(dev).train()
inp, gt = bench.make_inputs(dev, 0)
net = model.implicit_network.fold()[0]
ray_dirs, cam_loc = rend_util.get_camera_params(inp['uv'], inp['pose'], inp['intrinsics'])
om = torch.ones(ray_dirs.shape[0] * ray_dirs.shape[1], dtype=torch.bool, device=dev)
rt = model.ray_tracer
print('conf line_step_iters', rt.line_step_iters)
for ls in (0, 1, 2, 3, 5):
    rt.line_step_iters = ls
    ev = []
    rt.events = ev
    for _ in range(4):
        pts, mask, dists = rt(sdf=NativeSDF(net), cam_loc=cam_loc, object_mask=om, ray_directions=ray_dirs)
    torch.cuda.synchronize()
    c = rt.last_counters.cpu().tolist()
    e = ev[-1]
    print(f'line_step_iters {ls}: sphere rows {c[0]:6d}  sampler rays {c[5]:5d}  min-sdf rays {c[6]:5d}  hit {int(mask.sum()):5d}  sphere kernel {e[0].elapsed_time(e[1]) * 1e3:7.1f} us')
