"""Sanity + timing sweep of the training step over shapes / widths / dtypes / phases on one GPU (dev tool): every configuration must run,
give finite losses and gradients; prints ms/step and rays/s."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.model.loss import IDRLoss
from mvsdf_amd.optim import FlatAdam
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict
dev = torch.device('cuda', 0)
t = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items()}
cfgs = [  # W, B, P, V, tp, dtype
    (256, 8, 256, 4, 0.3, 'f32'), (256, 8, 256, 4, 0.05, 'f32'), (256, 8, 256, 4, 0.7, 'bf16x2'), (256, 1, 2048, 4, 0.3, 'f32'),
    (256, 8, 1024, 8, 0.3, 'f32'), (256, 8, 2048, 8, 0.3, 'bf16x2'), (256, 8, 4096, 8, 0.3, 'f32x3'), (512, 8, 256, 2, 0.3, 'f32'),
    (512, 8, 512, 2, 0.3, 'f32x3'), (64, 3, 37, 2, 0.3, 'f32'), (128, 2, 1000, 3, 0.3, 'f32x3'), (384, 2, 700, 3, 0.3, 'f32x3'), (320, 4, 2100, 2, 0.3, 'f32x3'),
]
for (W, B, P, V, tp, dt) in cfgs:
    model = IDRNetwork(ConfigDict(synth.model_conf(W)))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0).items()})
    model = model.to(dev).train().set_trace_dtype(dt)
    loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
    inp, gt = synth.make_batch(B, P, V, seed=1, feat_hw=(150, 200))
    if tp < 1 / 6:
        inp['depths'] = gt['depths'] = synth.make_depth_maps(inp['depth_cams'], 2.0, (0.0, 0.0, 0.0), seed=1, hole_frac=0.05)
    inp, gt = t(inp), t(gt)
    def step():
        opt.zero_grad(); out = model(inp, tp); lo = loss_fn(out, dict(gt), tp, B); opt.backward(lo['loss']); opt.step(grad_cap=2.0); return out, lo
    torch.manual_seed(0)
    for _ in range(3): out, lo = step()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 15
    for _ in range(n): out, lo = step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    ok = all(torch.isfinite(v).all() for v in lo.values()) and torch.isfinite(opt.flat_g).all() and float(opt.grad_norm()) > 0
    print('W=%d B=%d P=%d V=%d tp=%.2f %s: %7.2f ms/step %8.0f k rays/s  N=%d loss=%.4f |g|=%.3g %s' % (W, B, P, V, tp, dt, ms, B * P / ms, model.last_stats['N'], float(lo['loss']), float(opt.grad_norm()), 'ok' if ok else 'NOT FINITE'), flush=True)
    del model, opt, inp, gt
    torch.cuda.empty_cache()
