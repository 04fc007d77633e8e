#!/usr/bin/env python3
"""Time the REFERENCE itself (jzhangbs/MVSDF @ /root/reference, PyTorch on CPU) on the bench workload: one full training step
(IDRNetwork.forward + IDRLoss + backward + grad-norm + clip + Adam, idr_train.py:283-302) on the c2 batch of bench.py
(8 views x 256 px = 2048 rays, 4 source views, 8x256 / 4x256 networks), protocol of SURVEY.md 8(d): 1 warm-up step, median of 3, at 1
thread (the reference's own setting, idr_train.py:21) and at all cores of this container.

Runs ONLY where /root/reference exists (the build container); writes profiles/reference_cpu.json, which bench.py quotes in
`cpu_baseline.sample` (the reference cannot travel to the GPU box, so its number is measured here and carried as data).

    python tools/time_reference_cpu.py [--width 256] [--views 8] [--px 256] [--src 4]
"""
import argparse
import importlib.util
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--width', type=int, default=256)
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--px', type=int, default=256)
    ap.add_argument('--src', type=int, default=4)
    ap.add_argument('--tp', type=float, default=0.3)
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)                   # the import shim of the golden generator (stubs imageio / skimage / cv2, Tensor.cuda -> identity)
    spec.loader.exec_module(mg)
    from mvsdf_amd.utils import synth
    from model import conf as rconf                              # the reference's schedule module (on sys.path through the shim)
    m, _ = mg.build_model(a.width, 0)
    inp, gt = synth.make_batch(a.views, a.px, a.src, seed=0, feat_hw=(150, 200))      # same scene as bench.py (smaller feature maps: their size does not enter the cost)
    mi, gtt = {k: mg.T(v) for k, v in inp.items()}, {k: mg.T(v) for k, v in gt.items()}
    loss_fn = mg.IDRLoss()
    opt = torch.optim.Adam(m.parameters(), lr=0.0)
    m.train()
    R = a.views * a.px

    def step():
        t0 = time.perf_counter()
        opt.zero_grad()
        with mg.quiet():
            out = m(mi, a.tp)
            lo = loss_fn(out, dict(gtt), a.tp, a.views)
        lo['loss'].backward()
        torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).norm()
        if rconf.phase[0] <= a.tp and rconf.enable_grad_cap:
            torch.nn.utils.clip_grad_norm_(m.parameters(), rconf.grad_cap(a.tp))
        opt.step()
        return time.perf_counter() - t0

    res = {}
    for threads in (1, os.cpu_count()):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        step()                                                   # warm-up
        ts = sorted(step() for _ in range(3))
        res[threads] = ts[1]
        print('%3d thread(s): %.3f s per step = %.0f rays/s (runs: %s)' % (threads, ts[1], R / ts[1], ' '.join('%.3f' % t for t in ts)), flush=True)
    best = min(res, key=res.get)
    # ... and the CPU PORT (bench.py::cpu_port_step = what `cpu_baseline.value` times on the GPU box's host) on the SAME batch in the SAME container, at 1 and at all
    # threads: with it the two hosts can be related -- reference(GPU box host) ~ reference(here) x port(GPU box host) / port(here)
    import bench
    from oracle import oracle as O
    from threadpoolctl import threadpool_limits
    bench.W = a.width
    port = {}
    for threads in (1, os.cpu_count()):
        O.set_num_threads(threads)
        with threadpool_limits(limits=threads):
            one_step, Rp = bench.cpu_port_step(a.src, a.views, a.px, a.width)
            one_step()
            ts = sorted(one_step()[0] for _ in range(3))
        port[threads] = ts[1]
        print('port, %3d thread(s): %.3f s per step = %.0f rays/s (runs: %s)' % (threads, ts[1], Rp / ts[1], ' '.join('%.3f' % t for t in ts)), flush=True)
    out = {
        'what': 'reference PyTorch-CPU training step (forward + loss + backward + grad-norm + clip + Adam) on the bench batch',
        'workload': '%d views x %d px = %d rays, %d src views, 8x%d SDF MLP, train_progress %.2f' % (a.views, a.px, R, a.src, a.width, a.tp),
        'protocol': '1 warm-up step, median of 3 (SURVEY.md 8d)',
        'host': platform.processor() or platform.machine(), 'container_cores': os.cpu_count(), 'torch': torch.__version__,
        'seconds_per_step': {str(k): v for k, v in res.items()},
        'rays_per_s': {str(k): R / v for k, v in res.items()},
        'best_threads': best, 'best_rays_per_s': R / res[best],
        'port_seconds_per_step': {str(k): v for k, v in port.items()},
        'port_rays_per_s': {str(k): R / v for k, v in port.items()},
        'port_sample': 'bench.py::cpu_port_step on the same %d x %d rays, V = %d, W = %d: C oracle tracer (OpenMP) + numpy float64 differentiable half + feature loss with gradient; '
                       '1 warm-up, median of 3' % (a.views, a.px, a.src, a.width),
    }
    path = os.path.join(ROOT, 'profiles', 'reference_cpu.json')
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path)


if __name__ == '__main__':
    main()
