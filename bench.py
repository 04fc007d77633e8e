#!/usr/bin/env python3
"""bench.py -- traced rays/s for one MVSDF training step (forward + loss + backward + grad-norm/clip + Adam) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md section 8d "c2"): 8 views x 256 px = 2048 rays per GPU, 4 source views, 8x256 SDF
MLP + 4x256 rendering MLP, 10 sphere-tracing iterations, line_step_iters 3, 100 sampler steps, 8 secant steps, train_progress 0.3,
synthetic random-weight scene (mvsdf_amd.utils.synth), feature maps 32 x 600 x 800, weights frozen (Adam lr = 0, clip 2.0 included).
Weak scaling: every rank traces its own 2048 rays; one all-reduce on the flat gradient bucket per step.

Prints ONE JSON line (rank 0) incl. `roofline` (the dominant kernel: k_ray_samples, i.e. the tracing MLP on sampler / secant /
min-sdf rows; HIP events on the launch stream) and `cpu_baseline` (the CPU oracle on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork  # noqa: E402
from mvsdf_amd.model.loss import IDRLoss  # noqa: E402
from mvsdf_amd.optim import FlatAdam  # noqa: E402
from mvsdf_amd.utils import synth  # noqa: E402
from mvsdf_amd.utils.config import ConfigDict  # noqa: E402

W, B, P, V, TP = 256, 8, 256, 4, 0.3
FEAT_HW = (600, 800)
PEAK_F32_MFMA = 157.3          # TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32)


def flops_per_row(W):
    dims = synth.sdf_layer_dims(W)
    f_t = 2 * sum(i * o for i, o in dims[:-1]) + 2 * dims[-1][0]
    f_s = 2 * sum(i * o for i, o in dims)
    f_r = 2 * sum(i * o for i, o in synth.render_layer_dims(W))
    return f_t, f_s, f_r


def make_inputs(dev, seed):
    inp, gt = synth.make_batch(B, P, V, seed=seed, feat_hw=FEAT_HW, with_features=False)
    g = torch.Generator(device=dev).manual_seed(1234 + seed)
    base = torch.randn(1, 1, 32, 1, 1, generator=g, device=dev)
    f = torch.empty(B, 1 + V, 32, *FEAT_HW, device=dev)
    for b in range(B):                                         # per view: keeps every op below 2^31 elements at the larger configs
        noise = torch.randn(1 + V, 32, FEAT_HW[0] + 4, FEAT_HW[1] + 4, generator=g, device=dev)
        f[b] = torch.nn.functional.avg_pool2d(noise, 5, stride=1) * 2.4 + base[0]
    del noise
    to = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items()}
    inp, gt = to(inp), to(gt)
    gt['feat'] = f[:, 0].contiguous()
    gt['feat_src'] = f[:, 1:].contiguous()
    return inp, gt


def cpu_baseline(rays_per_view=None, views=None):
    """The CPU oracle (oracle/: C tracer with OpenMP + numpy float64 differentiable half) on a bounded sample of the workload
    (sized for roughly 10-30 s of CPU work: the whole 2048-ray batch on a many-core host, a quarter of it otherwise)."""
    from oracle import oracle as O
    from oracle import oracle_np as ON
    if rays_per_view is None:
        many = O.num_threads() >= 32
        views, rays_per_view = (B, P) if many else (2, 256)
    sd = synth.make_state_dict(W, 0)
    onet, nnet, rnet = O.Net(sd), ON.sdf_net(sd), ON.render_net(sd)
    inp, gt = synth.make_batch(views, rays_per_view, V, seed=0, feat_hw=(150, 200))
    tr = synth.model_conf(W)['ray_tracer']
    R = views * rays_per_view
    def one_step():
        rs = np.random.RandomState(0)
        t0 = time.time()
        dirs, cam = O.camera_rays(inp['uv'], inp['pose'], inp['intrinsics'])
        pts, mask, dists, rows = O.trace(onet, cam, dirs, np.ones(R, bool), True, rs.uniform(size=100).astype(np.float32), None, **tr)
        hit = np.nonzero(mask)[0]
        N, E = len(hit), R // 2
        x_all = np.concatenate([pts[hit], rs.uniform(-1, 1, size=(E, 3)), pts[~mask]], 0)
        y, n, cache = ON.sdf_forward(nnet, x_all)
        view = -dirs.reshape(-1, 3)[hit]
        rgb, rc = ON.render_forward(rnet, x_all[:N], n[:N], view, y[:N, 2:])
        counts = mask.reshape(views, -1).sum(1)
        ON.feat_corr_loss(x_all[:N], counts, gt['feat'], gt['cam'], gt['feat_src'], gt['src_cams'], gt['size'][0], gt['center'][0])
        dW, db, dp, dn_r, df = ON.render_backward(rnet, rc, rs.normal(size=rgb.shape))
        dy = np.zeros_like(y)
        dy[:N, 2:] = df
        dn = np.zeros((x_all.shape[0], 3))
        dn[:N] = dn_r
        dn[:N + E] += 2 * (np.linalg.norm(n[:N + E], axis=1, keepdims=True) - 1) * n[:N + E] / np.linalg.norm(n[:N + E], axis=1, keepdims=True) / (N + E)
        ON.sdf_backward(nnet, cache, dy, dn)
        return time.time() - t0, rows
    # protocol of SURVEY 8(d): one warm-up step, then the median of three
    one_step()
    runs = [one_step() for _ in range(3)]
    dt = sorted(r[0] for r in runs)[1]
    rows = runs[0][1]
    return {'value': R / dt, 'unit': 'rays/s', 'cores': O.num_threads(), 'kind': 'port',
            'sample': '%d views x %d rays of the same scene (W=%d, V=%d): C oracle tracer (OpenMP, %d rows) + numpy float64 value/normal fwd+bwd, '
                      'rendering fwd+bwd, feature loss fwd (its gradient omitted); median of 3 steps after 1 warm-up, %.1f s per step' % (views, rays_per_view, W, V, int(rows.sum()), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)          # ~3 ms per step: the default run still takes seconds
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        local %= max(1, torch.cuda.device_count())             # (a box with fewer GPUs than ranks: dry runs of the launch path only)
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get('MVSDF_DIST_BACKEND', 'nccl'))      # nccl = RCCL over xGMI; gloo for dry runs on one GPU
    elif a.gpus > 1:
        sys.exit('launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...')
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    model = IDRNetwork(ConfigDict(synth.model_conf(W)))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0).items()})
    model = model.to(dev).train()
    loss_fn = IDRLoss()
    # frozen weights (lr = 0): a step on random GT collapses the scene.  Parameters, gradients and Adam moments
    # live in flat buffers: one memset, one all-reduce, two launches for grad-norm + clip + Adam.
    opt = FlatAdam(model.parameters(), lr=0.0)
    inp, gt = make_inputs(dev, seed=rank)
    events = []
    model.ray_tracer.events = events

    def step():
        opt.zero_grad()
        out = model(inp, TP)
        lo = loss_fn(out, dict(gt), TP, B)
        lo['loss'].backward()
        opt.all_reduce_mean()
        opt.step(grad_cap=2.0)                                   # grad-norm + clip + Adam (idr_train.py:289-302, conf.grad_cap)
        return lo

    torch.manual_seed(rank)
    for _ in range(a.warmup):
        step()
    events.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        lo = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        R = B * P
        f_t, f_s, f_r = flops_per_row(W)
        st = model.last_stats
        cnt = st['counters'].cpu().numpy()
        # tracer rows actually evaluated: counters[8] = ray-sampler rows up to each ray's first sign change (the reference also
        # evaluates the samples behind it, counters[1], which no output reads)
        T, T_ref, N, E = int(cnt[0] + cnt[8] + cnt[2] + cnt[3]), int(cnt[:4].sum()), st['N'], R // 2
        flops_step = T * f_t + ((R + E) + 2 * (N + E)) * f_s + 3 * (N + E) * f_t + 3 * N * f_r
        ms_sphere = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
        ms_samples = float(np.mean([e[1].elapsed_time(e[2]) + e[3].elapsed_time(e[4]) if len(e) == 5 else e[1].elapsed_time(e[2]) for e in events]))
        rows_samples = int(cnt[8] + cnt[2] + cnt[3])
        n_launch = 3                    # k_ray_samples per step: sampler rows (first window), sampler rows (open rays), secant || min-sdf rows
        ach = rows_samples * f_t / (ms_samples * 1e-3) / 1e12
        res = {
            'metric': 'traced rays/sec (fwd+bwd, 10 sphere iters, 4 src views)', 'value': world * R * a.steps / dt, 'unit': 'rays/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'DTU-scan24-shaped synthetic scene, %d rays/GPU (%d views x %d px), %d src views, 8x%d SDF MLP, full fwd+loss+bwd+clip+Adam(lr=0)'
                                   % (R, B, P, V, W), 'rays_per_gpu': R, 'src_views': V, 'sdf_width': W, 'train_progress': TP,
                       'feature_maps': '32x%dx%d' % FEAT_HW, 'parallelism': 'ray-sharded dp%d, one all-reduce on a flat grad bucket' % world},
            'roofline': {'bound': 'mfma', 'kernel': 'k_ray_samples (fused 9-layer tracing MLP on the sampler / secant / min-sdf rows)', 'achieved': ach,
                         'peak': PEAK_F32_MFMA, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA,
                         # HBM bytes per launch from PMC passes of this command (profiles/r01_pmc_summary.csv), mean of the three launches:
                         # 2 x FETCH_SIZE (gfx950 correction for wide coalesced reads; FETCH_SIZE in KB = 1024 B) + WRITE_SIZE.  Not
                         # measurable from inside the process.  ~8.6x the 1.84 MB weight set: each of the 8 XCD L2s pulls its own copy.
                         'traffic': (2 * 7678.5 + 100.8) * 1024 if (W, B, P) == (256, 8, 256) else None,
                         'rows_per_launch': rows_samples / n_launch, 'flop_per_row': f_t, 'avg_launch_ms': ms_samples / n_launch,
                         'launches_per_step': n_launch,
                         'k_sphere_trace': {'rows_per_launch': int(cnt[0]), 'avg_launch_ms': ms_sphere,
                                            'achieved': int(cnt[0]) * f_t / (ms_sphere * 1e-3) / 1e12},
                         'step': {'T_trace_rows': T, 'T_reference_rows': T_ref, 'R': R, 'E': E, 'N_hit': N, 'flops_step': flops_step,
                                  'achieved': flops_step / (dt / a.steps) / 1e12, 'frac': flops_step / (dt / a.steps) / 1e12 / PEAK_F32_MFMA}},
            'loss': float(lo['loss'].detach()),
        }
        if not a.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
