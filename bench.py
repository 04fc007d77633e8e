#!/usr/bin/env python3
"""bench.py -- traced rays/s for one MVSDF training step (forward + loss + backward + grad-norm/clip + Adam) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5share] [--dtype f32|f32x3|bf16|bf16w|bf16x2|bf16x3] [--width 256|512]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process (which never touches the GPU) starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a child, relays
rank 0's JSON line and exits with the child's code.  Launched under torch.distributed.run directly it just runs as one rank.

Workload (BASELINE.json configs[1] = SURVEY.md 8d "c2" at N = 1, configs[3] = "c4" at N = 8): B = 8 views in total, 256 * N pixels per
view, the views sharded over the N ranks (parallel.shard_views) -> 2048 rays per GPU at every N (weak scaling: 2048 rays at N = 1,
16384 rays = 8 views x 2048 px, one view per GPU, at N = 8).  The depth maps of all 8 views stay on every rank (the depth term carves each
sample point against every view, loss.py:39-40); feature maps only of a rank's own views.  4 source views, 8x256 SDF MLP + 4x256 rendering
MLP, 10 sphere-tracing iterations, line_step_iters 3, 100 sampler steps, 8 secant steps, train_progress 0.3, synthetic random-weight
scene (mvsdf_amd.utils.synth), feature maps 32 x 600 x 800 stored channels-last (each bilinear tap one 128-byte line), weights frozen
(Adam lr = 0, clip 2.0 included).  Per step and rank: ONE all-reduce(SUM) of the flat fp32 gradient buffer (RCCL over xGMI), its 1/N folded
into the Adam launch, plus a 3-float all-reduce of the loss normaliser counts (IDRLoss.exact_data_parallel) that overlaps the forward.

The timed loop is the reference's training iteration without its per-step print (idr_train.py:253-315): zero_grad, forward, loss, backward, gradient
all-reduce, clip + Adam.  With the deferred step (IDRNetwork.deferred_step, the default) nothing in it waits for the GPU: the host enqueues whole steps ahead.

Prints ONE JSON line (rank 0) incl. `roofline` (the DOMINANT tracing-MLP kernel of the run -- k_sphere_trace or k_ray_samples, whichever took longer per step --
with both under `kernels`, HIP events on the launch stream; k_feat_corr's gather rate), `timing` (gpu_ms_per_step: HIP events around the timed region;
kernel_ms_per_step: event distance around each C call of a step; gpu_idle_frac; host milliseconds per section) and `cpu_baseline` (the CPU oracle on a bounded
sample of the same workload).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mvsdf_amd import ops  # noqa: E402
from mvsdf_amd.model.implicit_differentiable_renderer import IDRNetwork  # noqa: E402
from mvsdf_amd.model.loss import IDRLoss  # noqa: E402
from mvsdf_amd.optim import FlatAdam  # noqa: E402
from mvsdf_amd.parallel import shard_views  # noqa: E402
from mvsdf_amd.utils import synth  # noqa: E402
from mvsdf_amd.utils.config import ConfigDict  # noqa: E402

W, B, TP = 256, 8, 0.3
WORKLOADS = {                     # name: (pixels per view and GPU-count unit, source views)
    'c2': (256, 4),               # BASELINE configs[1]: 2048 rays, 4 source views (the configuration the metric is quoted on)
    'c3': (1024, 8),              # configs[2]: 8192 rays, 8 source views
    'c5share': (512, 8),          # configs[4] per-GPU share: 32768 rays / 8 GPUs = 4096 rays, 8 source views (meant for --dtype bf16x2)
    # the reference's REAL default training step on one GPU (not a BASELINE config; secondary line only): 8 views x 4096 px = 32768 rays (README.md:38 batch size,
    # confs/mvsdf_dtu.conf:4 num_pixels), 8x512 SDF net / 4x512 rendering net (mvsdf_dtu.conf:24,35), num_src = 2 source views (datasets/scene_dataset.py:104)
    'shipped': (4096, 2),
}
WORKLOAD_WIDTH = {'shipped': 512}
FEAT_HW = (600, 800)
P, V = WORKLOADS['c2']            # defaults of make_inputs (dev tools under tools/ set bench.B / bench.P / bench.V and call it)
PEAK = {'f32x3': 2500.0, 'f32': 157.3, 'bf16x2': 2500.0, 'bf16w': 157.3}    # dense MFMA TFLOP/s of the tracing MLP's matrix instruction,
# /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x32_bf16); bf16x2 issues 2 bf16 matrix instructions per
# ALGORITHMIC multiply-add (activations as bf16 terms), f32x3 six (fp32 weights as three terms too): `achieved` counts the algorithmic FLOPs once and
# `roofline.peak` is the instruction's dense peak divided by MUL (the rate at which the matrix pipe can deliver ALGORITHMIC multiply-adds in that arithmetic)
MUL = {'bf16x2': 2, 'f32x3': 6}


def reference_cpu():
    """The reference itself on CPU (PyTorch + MKL): it cannot travel to the GPU box, so it is timed in the build container by
    tools/time_reference_cpu.py (the golden generator's import shim + the protocol of SURVEY 8d) and carried as data in profiles/reference_cpu.json.
    -> structured record for cpu_baseline.reference (None when the file is missing)."""
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'reference_cpu.json')))
        threads = sorted(d['rays_per_s'], key=int)
        return {'what': d['what'], 'workload': d['workload'], 'protocol': d['protocol'], 'host': 'build container, %d cores (%s)' % (d['container_cores'], d['host']),
                'rays_per_s_1t': d['rays_per_s'].get('1'), 'rays_per_s_all': d['rays_per_s'][threads[-1]], 'threads': int(threads[-1]),
                'seconds_per_step_all': d['seconds_per_step'][threads[-1]], 'source': 'profiles/reference_cpu.json (tools/time_reference_cpu.py)',
                # the CPU port (cpu_port_step: what `cpu_baseline.value` times on this host) timed in the SAME container as the reference
                'port_rays_per_s_container': d.get('port_rays_per_s'), 'port_sample_container': d.get('port_sample')}
    except (OSError, ValueError, KeyError):
        return None


def flops_per_row(W):
    dims = synth.sdf_layer_dims(W)
    f_t = 2 * sum(i * o for i, o in dims[:-1]) + 2 * dims[-1][0]
    f_s = 2 * sum(i * o for i, o in dims)
    f_r = 2 * sum(i * o for i, o in synth.render_layer_dims(W))
    return f_t, f_s, f_r


def make_inputs(dev, rank=0, world=1, P=None, V=None):
    """This rank's share of the global batch (B views x P px, seed 0 on every rank): its views' rays / GT / feature maps, all depth maps."""
    P = globals()['P'] if P is None else P
    V = globals()['V'] if V is None else V
    inp, gt = synth.make_batch(B, P, V, seed=0, feat_hw=FEAT_HW, with_features=False)
    to = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items()}
    inp, gt = shard_views(to(inp), rank, world), shard_views(to(gt), rank, world)
    per = B // world
    H, Wd = FEAT_HW
    # channels-last storage [view, 1+V, H, W, C]: the features are constants of the dataset (scene_dataset.py:139-149 caches them once),
    # laid out so that each of the 4 bilinear taps of k_feat_corr is one 128-byte line instead of 32 scattered dwords
    store = torch.empty(per, 1 + V, H, Wd, 32, device=dev)
    base = torch.randn(1, 32, 1, 1, generator=torch.Generator(device=dev).manual_seed(1234), device=dev)
    for i in range(per):
        g = torch.Generator(device=dev).manual_seed(5000 + rank * per + i)          # per global view: independent of the sharding
        noise = torch.randn(1 + V, 32, H + 4, Wd + 4, generator=g, device=dev)
        store[i].copy_((torch.nn.functional.avg_pool2d(noise, 5, stride=1) * 2.4 + base).permute(0, 2, 3, 1))
    del noise
    gt['feat'] = store[:, 0].permute(0, 3, 1, 2)                                  # [b, C, H, W] view with stride(C) = 1
    gt['feat_src'] = store[:, 1:].permute(0, 1, 4, 2, 3)                          # [b, V, C, H, W]
    return inp, gt


def cpu_port_step(V, views, rays_per_view, width=None):
    """-> (one_step() -> (seconds, tracer rows), R): one training step of the CPU PORT (oracle/: C tracer with OpenMP = the fmaf-chain restatement of
    ray_tracing.py:27-98 + numpy float64 value / normal forward and double backward, rendering net forward / backward, feature-consistency loss WITH its
    analytic gradient) on `views` x `rays_per_view` rays of the bench scene.  Used by cpu_baseline() below (GPU box host) and by tools/time_reference_cpu.py
    (build container, beside the reference itself) -- the same function, so the two hosts can be related.  Since round 6 the step is complete: the weight-norm
    fold backward of both networks, the all-parameter gradient norm, clip_grad_norm_(2.0) and one Adam update (idr_train.py:289-302) on the
    float64 parameters (3 MB of elementwise work); still not in it: the eikonal / depth / surface / rgb loss terms' own arithmetic beyond their gradients' stand-ins
    (a random upstream on rgb, the eikonal term's exact gradient)."""
    from oracle import oracle as O
    from oracle import oracle_np as ON
    width = width or W
    sd = synth.make_state_dict(width, 0)
    onet, nnet, rnet = O.Net(sd), ON.sdf_net(sd), ON.render_net(sd)
    inp, gt = synth.make_batch(views, rays_per_view, V, seed=0, feat_hw=(150, 200))
    tr = synth.model_conf(width)['ray_tracer']
    R = views * rays_per_view
    adam = {}                                                    # Adam moments of the port's parameters (float64), by (network, layer, kind)

    def optimiser_tail(grads):
        # grads: [(net, dW list, db list)] -> fold backward into (v, g), gradient norm over every parameter, clip at 2.0, one Adam step (lr = 0: frozen weights)
        flat = []
        for net, dW, db in grads:
            for l in range(net.n_layers):
                dv, dg = ON.fold_backward(net.v[l], net.g[l], dW[l])
                flat += [(net, l, 'v', dv), (net, l, 'g', dg.reshape(net.g[l].shape)), (net, l, 'b', db[l])]
        norm = float(np.sqrt(sum(float((g * g).sum()) for *_, g in flat)))
        coef = min(2.0 / (norm + 1e-6), 1.0)
        for net, l, kind, g in flat:
            g = g * coef
            m, v = adam.setdefault((id(net), l, kind), (np.zeros_like(g), np.zeros_like(g)))
            m += (g - m) * 0.1
            v *= 0.999
            v += 0.001 * g * g
            getattr(net, kind)[l] -= 0.0 * (m / 0.1) / (np.sqrt(v / 0.001) + 1e-8)          # lr = 0
        return norm

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
        ON.feat_corr_loss(x_all[:N], counts, gt['feat'], gt['cam'], gt['feat_src'], gt['src_cams'], gt['size'][0], gt['center'][0], with_grad=True)
        dW_r, db_r, dp, dn_r, df = ON.render_backward(rnet, rc, rs.normal(size=rgb.shape))
        dy = np.zeros_like(y)
        dy[:N, 2:] = df
        dn = np.zeros((x_all.shape[0], 3))
        dn[:N] = dn_r
        dn[:N + E] += 2 * (np.linalg.norm(n[:N + E], axis=1, keepdims=True) - 1) * n[:N + E] / np.linalg.norm(n[:N + E], axis=1, keepdims=True) / (N + E)
        dW_s, db_s, _ = ON.sdf_backward(nnet, cache, dy, dn)
        optimiser_tail([(nnet, dW_s, db_s), (rnet, dW_r, db_r)])
        return time.time() - t0, rows
    return one_step, R


def cpu_baseline(V, rays_per_view=None, views=None):
    """The CPU port (cpu_port_step) on a bounded sample of the workload (sized for roughly 10-30 s of CPU work: the whole 2048-ray batch on a many-core
    host, a quarter of it otherwise), protocol of SURVEY 8(d): one warm-up step, then the median of three."""
    from oracle import oracle as O
    if rays_per_view is None:
        many = O.num_threads() >= 32
        views, rays_per_view = (B, 256) if many else (2, 256)
    one_step, R = cpu_port_step(V, views, rays_per_view)
    one_step()
    runs = [one_step() for _ in range(3)]
    dt = sorted(r[0] for r in runs)[1]
    rows = runs[0][1]
    ref = reference_cpu()
    if ref is not None and ref.get('port_rays_per_s_container'):
        # the reference cannot run on this host; the port ran on both: reference(this host) ~ reference(container) x port(this host) / port(container), all threads
        pc = ref['port_rays_per_s_container']
        k = max(pc, key=int)
        ref['reference_on_this_host_estimate_rays_per_s'] = ref['rays_per_s_all'] * (R / dt) / pc[k]
        ref['estimate_note'] = ('reference (container, %s threads) x port (this host, %d threads) / port (container, %s threads): a stated estimate -- the port is '
                                'OpenMP C + numpy, the reference PyTorch / MKL, their thread scaling differs' % (ref['threads'], O.num_threads(), k))
    return {'value': R / dt, 'unit': 'rays/s', 'cores': O.num_threads(), 'kind': 'port',
            'sample': '%d views x %d rays of the same scene (W=%d, V=%d): C oracle tracer (OpenMP, %d rows) + numpy float64 value/normal fwd+bwd, '
                      'rendering fwd+bwd, feature loss fwd + analytic gradient, weight-norm fold backward + gradient norm + clip + Adam; median of 3 steps after 1 warm-up, %.1f s per step'
                      % (views, rays_per_view, W, V, int(rows.sum()), dt),
            # the reference itself (PyTorch CPU, whole step incl. loss backward and Adam) -- measured where /root/reference exists, carried as data
            'reference': ref}


def self_launch(a):
    """--gpus N > 1 outside a launcher: start the N ranks as fresh child processes (this process has made no GPU call) and relay."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % a.gpus, '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


_PMC = None


def pmc_entry(workload, dtype, width, scaling='weak', rays=None):
    """The PMC record of this exact command (workload, tracing dtype, width, scaling mode, rays per GPU) from profiles/pmc_traffic.json, or None when no
    PMC pass of that command exists (a strong-scaling line must not carry the weak c2 step's traffic).  tools/pmc_to_json.py writes the file from separate
    --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 5 --warmup 2 --no-cpu-baseline [...]` (2 x FETCH_SIZE KB + WRITE_SIZE KB per the guide's
    gfx950 correction); the counters cannot be read from inside the process."""
    global _PMC
    if _PMC is None:
        try:
            _PMC = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        except (OSError, ValueError):
            _PMC = {}
    e = _PMC.get('entries', {}).get('%s|%s|%d|%s' % (workload, dtype, width, scaling))
    if e is not None and rays is not None and e.get('rays_per_gpu') not in (None, rays):
        return None
    return e


def pmc_traffic(entry, kernel):
    """HBM bytes per launch of `kernel` (mean over its launches, all template instances)."""
    k = None if entry is None else entry.get('kernels', {}).get(kernel)
    return None if k is None else k.get('hbm_bytes_per_launch')


def pmc_step_bytes(entry):
    """HBM bytes of one whole step: sum over the step's own kernels of bytes per launch x launches per step."""
    if entry is None:
        return None
    return sum(k['hbm_bytes_per_launch'] * k['launches_per_step'] for k in entry.get('kernels', {}).values())


def pin_rank_cpus(local, local_world):
    """Before any GPU call: give this rank its own slice of the host cores.  Every rank busy-polls a host-mapped sequence number once per step
    (mvsdf_step_wait_counts) -- unpinned, the 8 polling threads of a node wander over each other's cores."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = max(1, len(cpus) // max(1, local_world))
        mine = cpus[local * per:(local + 1) * per] or cpus
        os.sched_setaffinity(0, mine)
        return len(mine)
    except (AttributeError, OSError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)          # ~3 ms per step: the default run still takes seconds
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--dtype', default='f32x3', choices=sorted(PEAK), help="arithmetic of the no-grad tracing MLP (IDRNetwork.set_trace_dtype); the differentiable half computes in fp32 (SDF chains: three bf16 terms per value on the bf16 matrix cores, csrc/chain_x3.h)")
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'], help="--gpus N: 'weak' (default) keeps the rays per GPU fixed (views sharded, px per view x N: N = 8 is the c4 shape); "
                    "'strong' keeps the JOB fixed at the workload's N = 8 shape (c2: c4's 8 views x 2048 px = 16384 rays in total) and shards its views: N = 1 runs all of it on one GPU")
    ap.add_argument('--width', type=int, default=0, help='hidden width of both MLPs: 256 = BASELINE.json (8x256), 512 = the reference\'s shipped conf (mvsdf_dtu.conf:24,35)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shard', default='', help="R/W (tests): run as ONE process on the shard rank R of a W-rank job would get (its views, its pixels per view, its seed); no process group")
    ap.add_argument('--host-delay-us', type=float, default=0.0, help='diagnostic (never for a quoted number): busy-wait this many microseconds on the host after each of the six calls of a step '
                    '(zero_grad, forward, loss, backward, all-reduce, optimiser) -- a stand-in for a slow or contended host.  The deferred step absorbs it while the host stays faster than the GPU on '
                    'average; the classic step (MVSDF_DEFERRED_STEP=0) pays what lands behind its mid-step wait')
    ap.add_argument('--variants', action='store_true', help='also time the opt-in lazy_unused_outputs step (secondary number; off by default so that a profile of this command holds the headline step only)')
    a = ap.parse_args()
    global W
    W = a.width or WORKLOAD_WIDTH.get(a.workload, 256)
    a.width = W

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(a))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('MVSDF_DIST_BACKEND', 'nccl')       # nccl = RCCL over xGMI; gloo for dry runs of the launch path on one GPU
    if os.environ.get('MVSDF_BENCH_DRYRUN') == '1':              # launch-path check on a box without GPUs (tests/test_bench_launch.py):
        dist.init_process_group('gloo')                          # rendezvous + one collective + rank 0's line, no device work
        t_ = torch.tensor([float(rank + 1)])
        dist.all_reduce(t_)
        if rank == 0:
            print(json.dumps({'dry_run': True, 'n_gpus': world, 'sum_of_ranks_plus_1': float(t_), 'views_per_rank': B // world,
                              'px_per_view': WORKLOADS[a.workload][0] * world}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    cpus_pinned = pin_rank_cpus(local, local_world) if world > 1 else None      # (before the first GPU call)
    n_dev = torch.cuda.device_count()
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl' and n_dev < local_world:
            # RCCL needs one device per rank; sharing a device would only fail later, inside the first collective, with an opaque message
            sys.exit('bench.py --gpus %d: this node shows %d GPU(s) and the backend is nccl (RCCL): one device per rank is required '
                     '(MVSDF_DIST_BACKEND=gloo shares devices: launch-path tests only)' % (world, n_dev))
        local %= max(1, n_dev)                                   # (gloo dry runs on a box with fewer GPUs than ranks)
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    under_launcher = 'WORLD_SIZE' in os.environ
    if under_launcher:                                           # also at world size 1: the collectives then run through RCCL all the same
        if B % world:
            sys.exit('--gpus must divide the %d views of the batch' % B)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))

    P_unit, V = WORKLOADS[a.workload]
    srank, sworld = rank, world                                  # whose shard of how many (== rank / world unless --shard)
    if a.shard:
        assert world == 1, '--shard is for single-process runs'
        srank, sworld = (int(v) for v in a.shard.split('/'))
    # pixels per view.  weak: P_unit x ranks -- 2048 (c2) rays per GPU at every world size; strong: the 8-rank shape at every world size -- the job is fixed,
    # a rank gets B / world of its views (c2: 16384 rays in total = BASELINE configs[3]; one GPU runs all of them)
    P = P_unit * (B if a.scaling == 'strong' else sworld)
    per = B // sworld
    R = per * P                                                  # rays of this rank
    conf = synth.model_conf(W)
    model = IDRNetwork(ConfigDict(conf))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(W, 0).items()})
    model = model.to(dev).train()
    model.set_trace_dtype(a.dtype)                               # bf16*: bf16 weights in the tracing MLP (BASELINE configs[4])
    loss_fn = IDRLoss()
    # frozen weights (lr = 0): a step on random GT collapses the scene.  Parameters, gradients and Adam moments
    # live in flat buffers: one memset, one all-reduce, two launches for grad-norm + clip + Adam.
    opt = FlatAdam(model.parameters(), lr=0.0)
    inp, gt = make_inputs(dev, srank, sworld, P, V)

    grad_events = None                                           # set to a list during the extra steps: (start, end) events around the gradient all-reduce
    sect = None                                                  # set to a dict during the extra steps: host seconds per section + (start, end) events around the optimiser calls

    def spin():
        t_end = time.perf_counter() + a.host_delay_us * 1e-6
        while time.perf_counter() < t_end:
            pass

    def step():
        # idr_train.py:253-315 without its per-step print: with the deferred step (IDRNetwork.deferred_step, the default) nothing below waits for the GPU
        if sect is None and a.host_delay_us > 0:
            opt.zero_grad(); spin()
            out = model(inp, TP); spin()
            lo = loss_fn(out, dict(gt), TP, per); spin()
            opt.backward(lo['loss']); spin()
            opt.all_reduce_mean(defer_scale=True); spin()
            opt.step(grad_cap=2.0, zero_grad=True); spin()
            return out, lo
        if sect is None:
            opt.zero_grad()
            out = model(inp, TP)
            lo = loss_fn(out, dict(gt), TP, per)
            opt.backward(lo['loss'])                             # loss.backward() with the direct gradient sink (one launch for all dv/dg/db)
            opt.all_reduce_mean(defer_scale=True)                # ONE all-reduce(SUM) of the flat gradient buffer; / world inside Adam
            opt.step(grad_cap=2.0, zero_grad=True)               # grad-norm + clip + Adam (idr_train.py:289-302, conf.grad_cap); leaves zeros in the gradients: the next
            return out, lo                                       # iteration's zero_grad() (idr_train.py:283) costs no launch
        # the same calls with the host's clock between them and HIP events (launch stream) around the optimiser's launches
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        t0 = time.perf_counter()
        ev[0].record(); opt.zero_grad(); ev[1].record()
        t1 = time.perf_counter()
        out = model(inp, TP)
        t2 = time.perf_counter()
        lo = loss_fn(out, dict(gt), TP, per)
        t3 = time.perf_counter()
        opt.backward(lo['loss'])
        t4 = time.perf_counter()
        ev[2].record(); opt.all_reduce_mean(defer_scale=True); ev[3].record()
        t5 = time.perf_counter()
        ev[4].record(); opt.step(grad_cap=2.0, zero_grad=True); ev[5].record()
        t6 = time.perf_counter()
        for k, v in (('zero_grad', t1 - t0), ('forward', t2 - t1), ('loss', t3 - t2), ('backward', t4 - t3), ('all_reduce', t5 - t4), ('optimizer', t6 - t5)):
            sect['host'][k] = sect['host'].get(k, 0.0) + v
        sect['events'].append(ev)
        if grad_events is not None:
            grad_events.append((ev[2], ev[3]))
        return out, lo

    torch.manual_seed(srank)                                     # ranks draw different eikonal points / min-sdf steps
    for _ in range(a.warmup):
        step()
    if under_launcher:
        dist.barrier()
    torch.cuda.synchronize()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    g0.record()                                                  # GPU-side clock of the same region (launch stream): first launch of step 1 .. last launch of step K
    for _ in range(a.steps):
        out, lo = step()
    g1.record()
    t_host = time.perf_counter() - t0                            # the host is done enqueueing here; the GPU may still be steps behind (deferred step)
    if under_launcher:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_ms = g0.elapsed_time(g1) / a.steps
    ranks = None
    if under_launcher:
        mine = torch.tensor([dt, float(local), gpu_ms, t_host / a.steps * 1e3], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dts = [float(t_[0]) for t_ in allr]
        # per rank: wall, GPU-side time of the timed region (events on the launch stream) and the host's enqueue loop, ms per step -- a rank whose host
        # cannot keep up shows host_enqueue close to wall (its GPU waits for launches); with the deferred step the enqueue loop is well below the GPU time
        ranks = {'ms_min': min(dts) / a.steps * 1e3, 'ms_max': max(dts) / a.steps * 1e3, 'devices': len({int(t_[1]) for t_ in allr}),
                 'cpus_per_rank': cpus_pinned, 'wall_ms_per_step': [d_ / a.steps * 1e3 for d_ in dts],
                 'gpu_ms_per_step': [float(t_[2]) for t_ in allr], 'host_enqueue_ms_per_step': [float(t_[3]) for t_ in allr]}
        dt = max(dts)                                            # the MAX over the ranks is the job's time

    # (every rank runs these extra steps: step() holds the gradient collective)
    # per-kernel durations of the tracer: HIP events on the launch stream around k_sphere_trace, the sampler launches and the secant /
    # min-sdf launch, over 20 more steps of the same loop (recorded inside the native step driver, mvsdf_step_set_timing; outside the
    # timed region so that the headline number carries no event overhead)
    nt = 20
    st_native = getattr(model, '_last_step', None) if model.native_step else None
    tms = []
    if under_launcher:                                           # ... and of the two collectives (HIP events on the compute stream around each call)
        grad_events = []
        loss_fn.collective_events = []
    from mvsdf_amd import native_step as NS
    sect = {'host': {}, 'events': []}
    NS.loss_timing = []
    if st_native is not None:
        st_native.set_timing(True)
        for _ in range(nt):
            step()
            torch.cuda.synchronize()
            tms.append(st_native.times())
        st_native.set_timing(False)
        ms_sphere = float(np.mean([t_[0] for t_ in tms]))
        ms_samples = float(np.mean([t_[1] + t_[2] for t_ in tms]))
        ms_diff_fwd = float(np.mean([t_[3] for t_ in tms]))
        ms_diff_bwd = float(np.mean([t_[4] for t_ in tms]))
    else:
        events = []
        model.ray_tracer.events = events
        for _ in range(nt):
            step()
        torch.cuda.synchronize()
        model.ray_tracer.events = None
        ms_diff_fwd = ms_diff_bwd = None
        ms_sphere = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
        ms_samples = float(np.mean([e[1].elapsed_time(e[2]) + e[3].elapsed_time(e[4]) if len(e) == 5 else e[1].elapsed_time(e[2]) for e in events]))
    # time of the step's own launches: HIP events on the launch stream around each C call (every call enqueues all its launches in tens of microseconds, so
    # the distance between its two events is kernel time) -- forward (fold .. output gather), IDRLoss, backward, gradient collective, optimiser, zero_grad
    torch.cuda.synchronize()
    evs = sect['events']
    loss_ms = [e0.elapsed_time(e1) for e0, e1 in NS.loss_timing]
    NS.loss_timing = None
    evm = lambda i, j: float(np.mean([e[i].elapsed_time(e[j]) for e in evs])) if evs else None
    host_ms = {k: v / max(1, len(evs)) * 1e3 for k, v in sect['host'].items()}
    kernel_parts = None
    if tms and evs:
        kernel_parts = {'forward': float(np.mean([t_[5] for t_ in tms])), 'loss': float(np.mean(loss_ms)) if loss_ms else None,
                        'backward': float(np.mean([t_[4] for t_ in tms])), 'all_reduce': evm(2, 3) if world > 1 else 0.0, 'optimizer': evm(4, 5), 'zero_grad': evm(0, 1)}
    sect = None
    collective_ms = None
    if under_launcher:
        torch.cuda.synchronize()
        ms = lambda evs: float(np.mean([a_.elapsed_time(b_) for a_, b_ in evs])) if evs else None
        collective_ms = {'backend': backend, 'world': world, 'grad_all_reduce': ms(grad_events), 'grad_bytes': int(opt.flat_g.numel() * 4),
                         'loss_counts_all_reduce': ms(loss_fn.collective_events),
                         'note': 'mean over %d steps outside the timed region; events on the compute stream around each call (the time the stream is held, incl. waiting for the slowest rank)' % nt}
        grad_events, loss_fn.collective_events = None, None
    # job-level invariants for the scaling tests: hits over all ranks (the hit masks do not depend on the sharding) and the norm of the rank-averaged gradient
    hits_total = int(model.last_stats['N'])
    grad_norm = float(opt.grad_norm())
    if under_launcher:
        hv = torch.tensor([float(hits_total)], device=dev, dtype=torch.float64)
        dist.all_reduce(hv)
        hits_total = int(hv.item())
    if rank == 0:
        f_t, f_s, f_r = flops_per_row(W)
        peak = PEAK[a.dtype] / MUL.get(a.dtype, 1)
        st = model.last_stats
        cnt = st['counters'].cpu().numpy()
        # tracer rows actually evaluated: counters[8] = ray-sampler rows up to each ray's first sign change (the reference also
        # evaluates the samples behind it, counters[1], which no output reads)
        T, T_ref, N, E = int(cnt[0] + cnt[8] + cnt[2] + cnt[3]), int(cnt[:4].sum()), st['N'], R // 2
        flops_step = T * f_t + ((R + E) + 2 * (N + E)) * f_s + 3 * (N + E) * f_t + 3 * N * f_r
        rows_tail = int(cnt[12])                 # min-sdf rows evaluated by finished workgroups INSIDE k_sphere_trace (tail filling): that kernel's work
        rows_samples = int(cnt[8] + cnt[2] + cnt[3]) - rows_tail
        rows_sphere = int(cnt[0]) + rows_tail
        n_launch = 3                    # k_ray_samples per step: sampler rows (first window), sampler rows (open rays), secant || min-sdf rows
        ach = rows_samples * f_t / (ms_samples * 1e-3) / 1e12
        ach_sphere = rows_sphere * f_t / (ms_sphere * 1e-3) / 1e12
        ach_both = (rows_samples + rows_sphere) * f_t / ((ms_samples + ms_sphere) * 1e-3) / 1e12
        # k_feat_corr, the one HBM-shaped kernel: 4 taps x 32 channels x 4 B = 512 B per (point, view), timed with HIP events on
        # the launch stream over 20 launches on the last step's surface points
        pts = out['diff_surf_pts'].detach()
        _, view_start, _ = ops.loss_prep(out['network_object_mask'], out['object_mask'], out['object_mask_true'], per)
        fargs = (pts, view_start, gt['feat'], gt['feat_src'], gt['cam'], gt['src_cams'], gt['size'][:1], gt['center'][:1])
        ops.feat_corr(*fargs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.feat_corr(*fargs)
        e1.record()
        torch.cuda.synchronize()
        ms_feat = e0.elapsed_time(e1) / 20
        feat_bytes = 512 * pts.shape[0] * (1 + V)
        total_R = world * R
        pmc = pmc_entry(a.workload, a.dtype, a.width, a.scaling, R) if world == 1 else None
        # the differentiable half (fp32 arithmetic; its SDF chains as three bf16 terms per value since round 5): value + normal forward of every evaluated row, rendering net, their backward incl. the second-order SDF
        # pass and the weight gradients = the formula's non-T terms; time = HIP events around the forward behind the tracer and around mvsdf_step_backward
        # (no bubbles on the stream: the distances are kernel time; at c3 / the c5 share the E sample rows run beside the tracer and are not in it)
        flops_diff = ((R + E) + 2 * (N + E)) * f_s + 3 * (N + E) * f_t + 3 * N * f_r
        diff_k = None
        if ms_diff_fwd is not None and ms_diff_bwd:
            ms_diff = ms_diff_fwd + ms_diff_bwd
            hb = None if pmc is None else sum(k['hbm_bytes_per_launch'] * k['launches_per_step'] for n_, k in pmc['kernels'].items()
                                              if n_ in ('k_chain_fwd', 'k_chain_bwd2', 'k_chain_fwd_x3', 'k_chain_bwd2_x3', 'k_delta_apply', 'k_wgrad_net', 'k_reduce_net', 'k_render_chain_fwd', 'k_render_chain_bwd',
                                                        'k_step_bwd_assemble', 'k_fold_bwd_net', 'k_step_outputs'))
            diff_k = {'bound': 'mfma (SDF chains: fp32 values as three bf16 terms on v_mfma_f32_16x16x32_bf16, csrc/chain_x3.h -- ceiling 2500 / 6 TF/s; rendering chains and weight '
                               'gradients: v_mfma_f32_16x16x4_f32; peak = the fp32 instruction, what the same arithmetic could reach without the term split)',
                      'flops_per_step': flops_diff, 'ms_per_step': ms_diff, 'ms_forward': ms_diff_fwd, 'ms_backward': ms_diff_bwd,
                      'achieved': flops_diff / (ms_diff * 1e-3) / 1e12, 'peak': PEAK['f32'], 'frac': flops_diff / (ms_diff * 1e-3) / 1e12 / PEAK['f32'],
                      'traffic': hb, 'traffic_GBps': None if hb is None else hb / (ms_diff * 1e-3) / 1e9}
        res = {
            'metric': 'traced rays/sec (fwd+bwd, 10 sphere iters, %d src views)' % V, 'value': total_R * a.steps / dt, 'unit': 'rays/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True,
            'scaling': a.scaling, 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': '%s: DTU-scan24-shaped synthetic scene, %d views x %d px = %d rays in total, %d rays/GPU (%d view(s) per GPU), %d src views, 8x%d SDF MLP, '
                                   'full fwd+loss+bwd+clip+Adam(lr=0)' % (a.workload if (world == 1 and a.scaling == 'weak') else a.workload + (' x%d (c4 shape at 8 GPUs)' % world if a.scaling == 'weak' else ' strong scaling: the 8-rank job (c4 shape for c2) on %d GPU(s)' % world),
                                                                          B, P, total_R, R, per, V, W),
                       'rays_per_gpu': R, 'rays_total': total_R, 'hits_total': hits_total, 'grad_norm_after_all_reduce': grad_norm, 'views_total': B, 'src_views': V, 'sdf_width': W, 'train_progress': TP,
                       'feature_maps': '32x%dx%d channels-last' % FEAT_HW,
                       'parallelism': 'views sharded over %d rank(s), depth maps replicated; one all-reduce(SUM) on the flat grad buffer (+ 3 loss counts)' % world},
            # the DOMINANT kernel of this run: whichever of the two tracing-MLP kernels took longer per step (k_sphere_trace: 1 launch, k_ray_samples: 3)
            'roofline': dict({'bound': 'mfma'}, **({
                             'kernel': 'k_sphere_trace (fused 9-layer tracing MLP inside the sphere-tracing state machine, ray_tracing.py:101-196)', 'achieved': ach_sphere,
                             'peak': peak, 'unit': 'TFLOP/s', 'frac': ach_sphere / peak, 'traffic': pmc_traffic(pmc, 'k_sphere_trace'),
                             'rows_per_launch': rows_sphere, 'flop_per_row': f_t, 'avg_launch_ms': ms_sphere, 'launches_per_step': 1} if ms_sphere >= ms_samples else {
                             'kernel': 'k_ray_samples (fused 9-layer tracing MLP on the sampler / secant / min-sdf rows)', 'achieved': ach,
                             'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': pmc_traffic(pmc, 'k_ray_samples'),
                             'rows_per_launch': rows_samples / n_launch, 'flop_per_row': f_t, 'avg_launch_ms': ms_samples / n_launch, 'launches_per_step': n_launch}),
                         # the two tracing-MLP kernels side by side (they take about the same time at this size)
                         kernels={
                             'k_ray_samples': {'rows_per_step': rows_samples, 'ms_per_step': ms_samples, 'achieved': ach, 'frac': ach / peak},
                             'tracing (k_ray_samples + k_sphere_trace)': {'rows_per_step': rows_samples + rows_sphere, 'ms_per_step': ms_samples + ms_sphere,
                                                                          'achieved': ach_both, 'frac': ach_both / peak},
                             'k_sphere_trace': {'rows_per_step': rows_sphere, 'rows_min_sdf_tail': rows_tail, 'ms_per_step': ms_sphere, 'achieved': ach_sphere, 'frac': ach_sphere / peak,
                                                'traffic': pmc_traffic(pmc, 'k_sphere_trace')},
                             'differentiable': diff_k,
                             'k_feat_corr': {'bound': 'hbm', 'points': int(pts.shape[0]), 'views_per_point': 1 + V, 'bytes': feat_bytes, 'ms': ms_feat,
                                             'achieved_GBps': feat_bytes / (ms_feat * 1e-3) / 1e9, 'peak_GBps': 8000.0,
                                             'traffic': pmc_traffic(pmc, 'k_feat_corr')}},
                         step={'T_trace_rows': T, 'T_reference_rows': T_ref, 'R': R, 'E': E, 'N_hit': N, 'flops_step': flops_step,
                               'achieved': flops_step / (dt / a.steps) / 1e12, 'frac': flops_step / (dt / a.steps) / 1e12 / peak,
                               'hbm_bytes': pmc_step_bytes(pmc), 'pmc_source': None if pmc is None else pmc.get('note')}),
            'loss': float(lo['loss'].detach()),
        }
        # Does the wall clock of the timed region equal the GPU's own time?  gpu_ms_per_step: two HIP events on the launch stream around the K timed steps.
        # kernel_ms_per_step: sum over one step's C calls of the event distance around each (measured over the 20 extra steps; inside a call the launches are
        # back to back).  gpu_idle_frac = 1 - kernel / gpu: the share of the timed region in which the stream had nothing to run (a host that cannot keep up).
        # host_ms: host clock per section of a step (enqueue cost; with the deferred step none of them waits for the GPU) and in the whole timed loop.
        kms = None
        if kernel_parts is not None and all(v is not None for v in kernel_parts.values()):
            kms = sum(kernel_parts.values())
        res['timing'] = {'gpu_ms_per_step': gpu_ms, 'kernel_ms_per_step': kms, 'gpu_idle_frac': None if kms is None else max(0.0, 1.0 - kms / gpu_ms),
                         'wall_over_kernel': None if kms is None else (dt / a.steps * 1e3) / kms, 'kernel_ms': kernel_parts,
                         'host_ms_per_step_enqueue_loop': t_host / a.steps * 1e3, 'host_ms': host_ms,
                         'deferred_step': bool(getattr(model, 'deferred_step', False) and st_native is not None and st_native.can_defer),
                         'host_delay_us_per_call': a.host_delay_us,
                         'note': 'kernel_ms: HIP events around each C call of a step, mean of %d steps after the timed region; gpu_ms: events around the timed region' % nt}
        mul = MUL.get(a.dtype)
        if mul:
            # the term engines issue `mul` bf16 matrix instructions per ALGORITHMIC multiply-add: `peak` above is the instruction's dense peak / mul
            res['roofline']['matrix_instructions_per_mac'] = mul
            res['roofline']['instruction_peak'] = PEAK[a.dtype]
            res['roofline']['frac_of_instruction_peak_counting_algorithmic_flops_once'] = ach / PEAK[a.dtype]
            if a.dtype == 'f32x3':
                res['roofline']['frac_of_fp32_mfma_peak'] = ach / PEAK['f32']           # against what the fp32 matrix instruction could deliver for the same arithmetic
        if collective_ms is not None:
            res['collective_ms'] = collective_ms
        if ranks is not None:
            res['ranks'] = ranks
        if world == 1 and a.variants:
            # secondary number, never `value`: the same step with the opt-in IDRNetwork.lazy_unused_outputs (the min-sdf points of non-hit rays,
            # which the training loop never reads, are evaluated only when `points` / `sdf_output` are read -- here: never)
            model.lazy_unused_outputs = True
            nv = max(10, a.steps // 2)
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nv):
                step()
            torch.cuda.synchronize()
            dtv = (time.perf_counter() - t0) / nv
            model.lazy_unused_outputs = False
            res['variants'] = {'lazy_unused_outputs': {'ms_per_step': dtv * 1e3, 'rays_per_s': R / dtv, 'steps': nv,
                                                       'rows_not_evaluated_per_step': int(cnt[3]),
                                                       'note': 'opt-in, default off; loss and gradients identical (tests/test_gpu_lazy.py); not the headline value'}}
        if not a.no_cpu_baseline and world == 1 and a.dtype == 'f32x3':
            # secondary number, never `value`: the same step with the tracing MLP as a k-ascending fmaf chain on the fp32 matrix instruction
            # (IDRNetwork.set_trace_dtype('f32'): the arithmetic `value` was quoted on in rounds 1-4; bit-exact against the fmaf-chain oracle).  Same run, same
            # inputs, same frozen weights: the hit masks of the two arithmetics are compared here too.  (Not in profile runs: they pass --no-cpu-baseline.)
            mask_def = out['network_object_mask'].clone()
            loss_def = float(lo['loss'].detach())
            model.set_trace_dtype('f32')
            for _ in range(max(5, a.warmup // 2)):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                out3, lo3 = step()
            torch.cuda.synchronize()
            dt3 = (time.perf_counter() - t0) / a.steps
            model.set_trace_dtype(a.dtype)
            res['trace_arithmetic_f32'] = {
                'ms_per_step': dt3 * 1e3, 'rays_per_s': R / dt3, 'steps': a.steps, 'value_speedup_over_it': dt3 / (dt / a.steps),
                'hit_masks_differing_from_value_run': int((out3['network_object_mask'] != mask_def).sum()), 'rays': R,
                'loss_f32x3': loss_def, 'loss_f32': float(lo3['loss'].detach()),
                'note': "IDRNetwork.set_trace_dtype('f32') / bench.py --dtype f32: the fmaf-chain arithmetic on v_mfma_f32_16x16x4_f32 (8 instructions of 32 cycles per "
                        "32-wide k-block; `value` runs the product default 'f32x3': the same fp32 arithmetic from 6 bf16 instructions of 16 cycles, reproduced bit for bit "
                        "by the oracle's model of that instruction at the full batch size, tests/test_gpu_f32x3.py; the losses differ by the eikonal / min-sdf draws of "
                        "the two loops too)"}
        if not a.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline(V)
        print(json.dumps(res), flush=True)
    if under_launcher:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
