"""Does the GPU run the first steps of a fresh process slower than the later ones (power state / clock ramp)?  Per-step GPU time (HIP events between the steps,
no synchronisation inside the loop) of the bench step from the very first one.  usage: python tools/clock_ramp_probe.py [c2|c3|c5share] [steps]"""
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
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device('cuda', 0)
model = IDRNetwork(ConfigDict(synth.model_conf(bench.W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
model = model.to(dev).train()
loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
P_, V_ = bench.WORKLOADS[wl]
inp, gt = bench.make_inputs(dev, 0, 1, P_, V_)
torch.cuda.synchronize()
time.sleep(float(os.environ.get('IDLE_S', '0')))                 # an idle GPU before the first step
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); opt.backward(lo['loss']); opt.step(grad_cap=2.0, zero_grad=True)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(wl, 'per-step GPU ms:', ' '.join('%.2f' % m for m in ms[:12]), '...', 'steps 20-29 mean %.3f' % (sum(ms[20:30]) / 10), 'steps 50-59 %.3f' % (sum(ms[50:60]) / 10), 'last 20 %.3f' % (sum(ms[-20:]) / 20))
# cumulative: what a benchmark of 20 steps after 5 warm-up steps would read, started at step s
for s in (0, 5, 30, 80):
    if s + 25 <= n:
        print('   20 steps after 5 warm-up, starting at step %3d: %.3f ms per step' % (s, sum(ms[s + 5:s + 25]) / 20))
# ... and again in the SAME process after an idle pause: a ramp that comes back is the GPU's power state, one that does not is one-time work of the process
for pause in (0.05, 1.0, 5.0):
    time.sleep(pause)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(26)]
    ev[0].record()
    for i in range(25):
        opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); opt.backward(lo['loss']); opt.step(grad_cap=2.0, zero_grad=True)
        ev[i + 1].record()
    torch.cuda.synchronize()
    m2 = [ev[i].elapsed_time(ev[i + 1]) for i in range(25)]
    print('   after %.2f s idle: %s | 20 steps after 5 warm-up: %.3f ms per step' % (pause, ' '.join('%.2f' % v for v in m2[:8]), sum(m2[5:]) / 20))
