"""Dev probe: the min-sdf row launch alone (mvsdf_trace_stage 5) against k_sdf_col0 on the same number of rows: what do the row gather, the
chunk prologues and the value stores of k_ray_samples cost on top of the bare MLP?"""
import ctypes as C
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from helpers import sdf_packed_net, trace_params
from mvsdf_amd import ops
from mvsdf_amd._lib import TraceParams, lib, ptr, stream_of, check
from mvsdf_amd.utils import synth

dev = torch.device('cuda', 0)
net = sdf_packed_net(synth.make_state_dict(256, 0))
if os.environ.get('TRACE_DTYPE', 'f32x3') == 'f32x3':               # the product default since round 5 (TRACE_DTYPE=f32: the fmaf-chain engine)
    net = ops.pack_bf16_net(net, terms=3, weight_terms=3)
inp, gt = bench.make_inputs(dev, 0)
dirs, cam = ops.camera_rays(inp['uv'], inp['pose'], inp['intrinsics'])
B, P = dirs.shape[:2]; R = B * P
om = torch.ones(R, dtype=torch.uint8, device=dev)
iv = torch.linspace(0, 1, 100, device=dev)
st = torch.rand(100, device=dev)
pts = torch.empty(R, 3, device=dev); mask = torch.empty(R, dtype=torch.uint8, device=dev); dists = torch.empty(R, device=dev)
counters = torch.empty(16, dtype=torch.int64, device=dev)
tp = TraceParams(*trace_params(256))
wsb = lib().mvsdf_trace_workspace_bytes_n(R, tp.n_steps)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
d = net.desc()
args = (C.byref(d), C.byref(tp), ptr(cam), ptr(dirs), ptr(om), B, P, 1, ptr(iv), ptr(st), ptr(pts), ptr(mask), ptr(dists), ptr(counters), ptr(ws),
        C.c_size_t(wsb), 1, int(os.environ.get('MT_SAMPLES', 4)), stream_of(dirs))   # (the step: mt 1, mt_samples 4)
def timed(stage, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        check(lib().mvsdf_trace_stage(1, *args), 's1'); check(lib().mvsdf_trace_stage(3, *args), 's3')
        if stage == 5: check(lib().mvsdf_trace_stage(6, *args), 's6')
        e0.record(); check(lib().mvsdf_trace_stage(stage, *args), 's'); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
t4 = timed(4)
cnt = counters.cpu().numpy()
t5, t6 = timed(5), timed(6)
rows_min, rows_sec = int(cnt[3]), int(cnt[2])
x = torch.rand(rows_min, 3, device=dev) * 2 - 1
for _ in range(3): ops.sdf_col0(net, x, mt=2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.sdf_col0(net, x, mt=2)
e1.record(); torch.cuda.synchronize()
tc = e0.elapsed_time(e1) / 20
print('min-sdf rows %d, secant rows %d' % (rows_min, rows_sec))
print('stage 4 (secant || rows + reduce) %.1f us; stage 5 (rows + reduce alone) %.1f us; stage 6 (secant alone) %.1f us; k_sdf_col0 on %d rows %.1f us' % (t4 * 1e3, t5 * 1e3, t6 * 1e3, rows_min, tc * 1e3))

# late round 6: the two halves of stage 4 as two launches on two streams (secant chains first, on a high-priority stream; a 64-row sample workgroup and a
# chain workgroup do not fit one CU's LDS together, so the chains keep their CUs to themselves) against the one launch that mixes them
def args_on(stream):
    return args[:-1] + (C.c_void_p(stream.cuda_stream),)
sA, sB = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
def timed_pair(n=20):
    cur = torch.cuda.current_stream()
    ts = []
    for _ in range(n):
        check(lib().mvsdf_trace_stage(1, *args), 's1'); check(lib().mvsdf_trace_stage(3, *args), 's3')
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eA, eB = torch.cuda.Event(), torch.cuda.Event()
        e0.record(cur); sA.wait_event(e0); sB.wait_event(e0)
        check(lib().mvsdf_trace_stage(6, *args_on(sA)), 's6'); eA.record(sA)
        check(lib().mvsdf_trace_stage(5, *args_on(sB)), 's5'); eB.record(sB)
        cur.wait_event(eA); cur.wait_event(eB); e1.record(cur); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
print('stage 6 || stage 5 on two streams: %.1f us' % (timed_pair() * 1e3))
