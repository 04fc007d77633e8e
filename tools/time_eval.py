"""Eval-mode rendering throughput (eval.py-style): one 600x800 view split into 10k-pixel chunks, IDR_RENDER (40 iterations) on."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['IDR_RENDER'] = '1'
import bench
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.general import split_input, merge_output
from mvsdf_amd.utils.config import ConfigDict
dev = torch.device('cuda', 0)
model = IDRNetwork(ConfigDict(synth.model_conf(bench.W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
model = model.to(dev).eval()
Wd, Hd = 800, 600
inp, _ = synth.make_batch(1, 8, 0, seed=0, feat_hw=(30, 40), with_features=False)
ys, xs = np.mgrid[0:Hd, 0:Wd]
inp['uv'] = np.stack([xs.reshape(-1), ys.reshape(-1)], -1)[None].astype(np.float32)
inp['object_mask'] = np.ones((1, Hd * Wd), dtype=bool)
inp = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
total = Hd * Wd
for chunk in (10000, 40000):
    def render():
        res = []
        for s in split_input(inp, total, n_pixels=chunk):
            with torch.no_grad():
                out = model(s)
            res.append({'rgb_values': out['rgb_values'].detach(), 'network_object_mask': out['network_object_mask'].detach()})
        return merge_output(res, total, 1)
    render(); torch.cuda.synchronize(); t0 = time.perf_counter()
    o = render(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'chunk {chunk}: {dt * 1e3:.1f} ms per {Wd}x{Hd} image = {total / dt / 1e6:.2f} Mrays/s, hit {float(o["network_object_mask"].float().mean()):.2f}')
