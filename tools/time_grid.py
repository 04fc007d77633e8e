"""SDF on a resolution^3 grid (mesh extraction, plots.py): time through the tracing-MLP kernel."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict
from mvsdf_amd.utils.plots import sdf_on_uniform_grid
m = IDRNetwork(ConfigDict(synth.model_conf(256)))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(256, 0).items()})
m = m.cuda().eval()
sdf = m.implicit_network.native_sdf()
for res in (100, 256, 512):
    sdf_on_uniform_grid(sdf, 32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z = sdf_on_uniform_grid(sdf, res)
    dt = time.perf_counter() - t0
    print(f'resolution {res}: {res ** 3 / 1e6:.1f} M points in {dt:.3f} s = {res ** 3 * 918016 / dt / 1e12:.1f} TFLOP/s incl. host copy')
