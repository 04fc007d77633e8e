"""Step time at the shipped conf's width (8x512 SDF MLP, 4x512 rendering MLP) on the bench batch."""
import os, sys, time
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
for W in (256, 512):
    model = IDRNetwork(ConfigDict(synth.model_conf(W)))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0).items()})
    model = model.to(dev).train()
    loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
    inp, gt = bench.make_inputs(dev, 0)
    def step():
        opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); lo['loss'].backward(); opt.step(grad_cap=2.0)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    c = model.last_stats['counters'].cpu().tolist()
    f_t, f_s, f_r = bench.flops_per_row(W)
    T = c[0] + c[8] + c[2] + c[3]
    print(f'W={W}: {dt * 1e3:.2f} ms/step  {bench.B * bench.P / dt / 1e3:.0f} k rays/s  tracer rows {T}  tracer-only FLOP rate if the tracer were the whole step: {T * f_t / dt / 1e12:.1f} TF')
