"""A/B: row tiles per chunk of the sample-row kernels (RayTracing.mt_samples) at the bench shapes, default tracing arithmetic.  python tools/ab_mt_samples.py [workload]"""
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
wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
P, V = bench.WORKLOADS[wl]
dev = torch.device('cuda', 0)
inp, gt = bench.make_inputs(dev, 0, 1, P, V)
for mts in (None, 1, 2, 4):
    for mt in (None,):
        model = IDRNetwork(ConfigDict(synth.model_conf(bench.W)))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
        model = model.to(dev).train()
        model.ray_tracer.mt_samples, model.ray_tracer.mt = mts, mt
        loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
        def step():
            opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); opt.backward(lo['loss']); opt.step(grad_cap=2.0)
        for _ in range(20): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(150): step()
        torch.cuda.synchronize()
        print('%s mt_samples=%s mt=%s: %.4f ms/step' % (wl, mts, mt, (time.perf_counter() - t0) / 150 * 1e3), flush=True)
