"""Step time of the other BASELINE configurations' shapes on one GPU (c3: 8192 rays, 8 source views; c4-shaped: 16384 rays)."""
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
for (B, P, V) in ((8, 256, 4), (8, 1024, 8), (16, 1024, 8), (32, 1024, 8)):
    bench.B, bench.P, bench.V = B, P, V
    model = IDRNetwork(ConfigDict(synth.model_conf(bench.W)))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
    model = model.to(dev).train()
    loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
    inp, gt = bench.make_inputs(dev, 0)
    for ms in (os.environ.get('MVSDF_MT_SAMPLES', '2'),):
        def step():
            opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, B); lo['loss'].backward(); opt.step(grad_cap=2.0)
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        c = model.last_stats['counters'].cpu().tolist()
        print(f'B={B} P={P} V={V}: {B * P} rays  {dt * 1e3:.2f} ms/step  {B * P / dt / 1e3:.0f} k rays/s   tracer rows {c[0] + c[8] + c[2] + c[3]}')
    del model, opt, inp, gt
    torch.cuda.empty_cache()
