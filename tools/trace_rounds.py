"""Per-round request history of the sphere-tracing state machine on the bench scene (generic emit / consume path with the HIP tracing
MLP as the callable): which ray sides ask for an evaluation in which round and phase.  Feeds the offline schedule model
(tools/sphere_schedule_model.py).  Writes gpurun_out/trace_rounds.npz."""
import os
import sys
import ctypes as C
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import sdf_packed_net, t, trace_params
from mvsdf_amd import ops
from mvsdf_amd._lib import TraceParams, lib, ptr, stream_of, check
from mvsdf_amd.utils import synth

W, B, P = 256, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 256
sd = synth.make_state_dict(W, 0)
net = sdf_packed_net(sd)
inp, _ = synth.make_batch(B, P, 0, seed=0, with_features=False)
dirs, cam = ops.camera_rays(t(inp['uv']), t(inp['pose']), t(inp['intrinsics']))
R = B * P
tp = TraceParams(*trace_params(W))
L, st = lib(), stream_of(dirs)
state = torch.empty(L.mvsdf_tracegen_state_bytes(R), dtype=torch.uint8, device='cuda')
req = torch.empty(R, 2, dtype=torch.uint8, device='cuda')
rpts = torch.empty(R, 2, 3, device='cuda')
vals = torch.zeros(R, 2, device='cuda')
counters = torch.empty(16, dtype=torch.int64, device='cuda')
om = torch.ones(R, dtype=torch.uint8, device='cuda')
check(L.mvsdf_tracegen_init(C.byref(tp), ptr(cam), ptr(dirs), ptr(om), B, P, ptr(state), ptr(req), ptr(rpts), ptr(counters), st))
hist_req, hist_phase, hist_k = [], [], []
while True:
    r = req.cpu().numpy().copy()
    if r.sum() == 0:
        break
    s = state.view(torch.int32).view(R, 16).cpu().numpy()
    hist_req.append(r); hist_phase.append(s[:, 12].copy()); hist_k.append(s[:, 11].copy())
    idx = torch.nonzero(req.view(-1)).flatten()
    vals.view(-1)[idx] = ops.sdf_col0(net, rpts.view(-1, 3)[idx].contiguous())
    check(L.mvsdf_tracegen_step(C.byref(tp), ptr(cam), ptr(dirs), B, P, ptr(state), ptr(vals), ptr(req), ptr(rpts), ptr(counters), st))
hist_req = np.stack(hist_req); hist_phase = np.stack(hist_phase); hist_k = np.stack(hist_k)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'trace_rounds_%d.npz' % R), req=hist_req, phase=hist_phase, k=hist_k)
print('rounds', hist_req.shape[0], 'rows', int(hist_req.sum()), 'counter', int(counters[0]))
