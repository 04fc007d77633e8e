"""How fast do rays leave the sphere-tracing loop?  Counters after k iterations (k = 1..10) on the bench workload."""
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
for k in range(1, 11):
    rt.sphere_tracing_iters = k
    ev = []
    rt.events = ev
    for _ in range(3):
        pts, mask, dists = rt(sdf=NativeSDF(net), cam_loc=cam_loc, object_mask=om, ray_directions=ray_dirs)
    torch.cuda.synchronize()
    c = rt.last_counters.cpu().tolist()
    e = ev[-1]
    print(f'iters {k:2d}: sphere rows {c[0]:6d}  unfinished (sampler) {c[5]:5d}  left-out (min-sdf) {c[6]:5d}  hit {int(mask.sum()):5d}   sphere kernel {e[0].elapsed_time(e[1]) * 1e3:7.1f} us')
