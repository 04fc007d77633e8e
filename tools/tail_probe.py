"""k_sphere_trace's tail: per-workgroup rounds / tracing time / helping time and how many min-sdf rows the finished workgroups served
(csrc/trace.hip::mv_tail_help).  MVSDF_TAIL=0|1 python tools/tail_probe.py [c2|c3|c5share] [bf16]   (the switch is read once per process)"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mvsdf_amd._lib import lib
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork
from mvsdf_amd.model.loss import IDRLoss
from mvsdf_amd.optim import FlatAdam
from mvsdf_amd.utils import synth
from mvsdf_amd.utils.config import ConfigDict
wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
dev = torch.device('cuda', 0)
model = IDRNetwork(ConfigDict(synth.model_conf(bench.W)))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(bench.W, 0).items()})
model = model.to(dev).train()
if 'bf16' in sys.argv: model.set_trace_dtype('bf16')
loss_fn = IDRLoss(); opt = FlatAdam(model.parameters(), lr=0.0)
P_, V_ = bench.WORKLOADS[wl]
inp, gt = bench.make_inputs(dev, 0, 1, P_, V_)
def step():
    opt.zero_grad(); out = model(inp, bench.TP); lo = loss_fn(out, dict(gt), bench.TP, bench.B); opt.backward(lo['loss']); opt.step(grad_cap=2.0)
for _ in range(10): step()
torch.cuda.synchronize()
nwg = 4096
probe = torch.zeros(nwg, 4, dtype=torch.int32, device=dev)
L = lib()
L.mv_trace_set_tail_probe.argtypes = [C.c_void_p]; L.mv_trace_set_tail_probe.restype = None
L.mv_trace_set_tail_probe(probe.data_ptr())
st = model._last_step
st.set_timing(True)
tms, rows = [], []
for _ in range(20):
    step(); torch.cuda.synchronize()
    tms.append(st.trace_times()); rows.append(model.last_stats['counters'].cpu().numpy().copy())
L.mv_trace_set_tail_probe(None)
p = probe.cpu().numpy().astype(np.int64)
used = p[:, 0] > 0
if not used.any(): used = p[:, 2] > 0
g = p[p[:, 2] > 0]
tms = np.array(tms); rows = np.array(rows)
print('workload %s, MVSDF_TAIL=%s: sphere %.1f us, sampler windows %.1f us, secant+minsdf %.1f us  (sum %.1f)' % (
    wl, os.environ.get('MVSDF_TAIL', '1'), *(tms.mean(0) * 1e3), tms.mean(0).sum() * 1e3))
print('min-sdf rows per step %d, of which inside k_sphere_trace %d; sphere rows %d' % (rows[:, 3].mean(), rows[:, 12].mean(), rows[:, 0].mean()))
if len(g):
    t_tr, t_help = g[:, 2] / 100.0, g[:, 3] / 100.0
    print('workgroups %d; rounds per workgroup: min %d median %d max %d' % (len(g), g[:, 0].min(), np.median(g[:, 0]), g[:, 0].max()))
    print('tracing time per workgroup (us): min %.0f median %.0f p90 %.0f max %.0f' % (t_tr.min(), np.median(t_tr), np.percentile(t_tr, 90), t_tr.max()))
    idle = (t_tr.max() - t_tr)
    print('CU time after a workgroup is done until the slowest is: mean %.0f us = %.1f %% of the kernel; units helped per workgroup mean %.2f max %d; helping time mean %.0f us max %.0f us' % (
        idle.mean(), 100 * idle.mean() / t_tr.max(), g[:, 1].mean(), g[:, 1].max(), t_help.mean(), t_help.max()))
    hist = np.bincount(g[:, 0].astype(int))
    print('workgroups by number of rounds:', {i: int(h) for i, h in enumerate(hist) if h})
