import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import sdf_packed_net, t
from mvsdf_amd import ops
from mvsdf_amd.utils import synth
sd = synth.make_state_dict(256, 0)
net = sdf_packed_net(sd)
M = 3072
x = torch.rand(M, 3, device='cuda') * 2 - 1
dy = torch.randn(M, 258, device='cuda') * 0.1; dn = torch.randn(M, 3, device='cuda')
y, n, ctx = ops.sdf_forward(net, x, M)
torch.cuda.synchronize()
def host_time(fn, reps=20):
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
    return np.median(ts) * 1e3
def gpu_time(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for name, fn in [('sdf_forward (19 launches)', lambda: ops.sdf_forward(net, x, M)),
                 ('sdf_backward full (fused chains)', lambda: ops.sdf_backward(net, x, M, M, 2550, dy[:2550], dn[:2550], ctx, True)),
                 ('sdf_backward dx only', lambda: ops.sdf_backward(net, x, M, M, 1523, dy[:1523], dn[:1523], ctx, True, want_dw=False)),
                 ('torch add tiny', lambda: x + 1.0), ('torch empty', lambda: torch.empty(10, device='cuda'))]:
    print('%-36s host %.3f ms   gpu %.3f ms' % (name, host_time(fn), gpu_time(fn)))
