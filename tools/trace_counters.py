"""Device-side tracer counters of the bench workload: rows / rays per tracer stage."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict
dev = torch.device('cuda', 0)
model = IDRNetwork(ConfigDict(synth.model_conf(bench.W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
model = model.to(dev).train()
inp, gt = bench.make_inputs(dev, 0)
out = model(inp, bench.TP)
c = model.last_stats['counters'].cpu().tolist()
names = ['ROWS_SPHERE', 'ROWS_SAMPLER', 'ROWS_SECANT', 'ROWS_MINSDF', 'N_SECANT', 'N_SAMPLER', 'N_MINSDF']
for n, v in zip(names, c): print(f'{n:14s} {v}')
print('N hit', model.last_stats['N'])
