"""`python bench.py --gpus N` must be runnable by the driver as is: with N > 1 and no launcher environment the parent process starts the
N ranks itself (torch.distributed.run on 127.0.0.1), relays rank 0's one JSON line and returns the children's exit code.  On this CPU box
the ranks run in MVSDF_BENCH_DRYRUN mode (gloo rendezvous + one collective, no device work); tests/test_gpu_dp.py and the gpu-marked
test below cover the real step."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env_extra, timeout=600):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None), env.pop('RANK', None), env.pop('LOCAL_RANK', None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    return p.returncode, lines, p.stderr.decode()[-3000:]


def test_self_launch_two_ranks_dry_run():
    rc, lines, err = _run(['--gpus', '2', '--steps', '2', '--warmup', '1'], {'MVSDF_BENCH_DRYRUN': '1'})
    assert rc == 0, err
    assert len(lines) == 1, (lines, err)
    d = json.loads(lines[0])
    assert d == {'dry_run': True, 'n_gpus': 2, 'sum_of_ranks_plus_1': 3.0, 'views_per_rank': 4, 'px_per_view': 512}


def test_self_launch_propagates_failure():
    rc, lines, err = _run(['--gpus', '3'], {'MVSDF_BENCH_DRYRUN': '0', 'CUDA_VISIBLE_DEVICES': ''})   # 3 does not divide 8 views / no GPU: children fail
    assert rc != 0 and not lines


def test_refuses_more_ranks_than_devices_under_rccl():
    """`--gpus N` with the nccl backend on a node that shows fewer than N devices: every rank exits with a message that names the cause (round 3 wrapped the
    device index instead and let RCCL fail inside its first collective)."""
    rc, lines, err = _run(['--gpus', '2', '--steps', '2', '--warmup', '1'], {'MVSDF_BENCH_DRYRUN': '0', 'CUDA_VISIBLE_DEVICES': ''})
    assert rc != 0 and not lines
    assert 'one device per rank is required' in err, err


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_gloo_full_step():
    """The real data-parallel step through the self-launcher: two ranks sharing the one GPU of the test box (gloo transport), B = 8 views x
    512 px sharded 4 + 4; the line reports n_gpus = 2 and 2 x 2048 rays per step."""
    rc, lines, err = _run(['--gpus', '2', '--steps', '3', '--warmup', '2', '--no-cpu-baseline'], {'MVSDF_DIST_BACKEND': 'gloo'})
    assert rc == 0, err
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 2 and d['config']['rays_per_gpu'] == 2048 and d['config']['rays_total'] == 4096
    assert d['value'] > 0 and d['roofline']['frac'] > 0 and d['scaling'] == 'weak'


@pytest.mark.gpu
def test_eight_ranks_on_one_gpu_gloo_c4_shape_and_collective_timings():
    """`bench.py --gpus 8` exactly as the driver calls it, with the eight ranks sharing the one GPU of the test box (gloo transport): the real
    c4 shape (8 views x 2048 px = 16384 rays, one view per rank); the line reports n_gpus = 8 and per-collective timings."""
    rc, lines, err = _run(['--gpus', '8', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], {'MVSDF_DIST_BACKEND': 'gloo', 'OMP_NUM_THREADS': '2'}, timeout=1200)
    assert rc == 0, err
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 8 and d['config']['rays_per_gpu'] == 2048 and d['config']['rays_total'] == 16384 and d['scaling'] == 'weak'
    c = d['collective_ms']
    assert c['world'] == 8 and c['backend'] == 'gloo' and c['grad_all_reduce'] > 0 and c['loss_counts_all_reduce'] > 0 and c['grad_bytes'] > 3e6
    print('8 ranks on one GPU (gloo): %.1f ms per step, gradient all-reduce %.2f ms, loss-count all-reduce %.2f ms' % (d['ms_per_step'], c['grad_all_reduce'], c['loss_counts_all_reduce']))
    # per rank: wall, GPU-side time of the timed region and the host's enqueue loop (the readiness record for a node whose ranks share host cores: a rank whose
    # host cannot keep up shows its enqueue loop at its wall time)
    r = d['ranks']
    assert len(r['wall_ms_per_step']) == len(r['gpu_ms_per_step']) == len(r['host_enqueue_ms_per_step']) == 8
    for k in range(8):
        print('   rank %d: wall %.2f ms, gpu %.2f ms, host enqueue loop %.2f ms per step' % (k, r['wall_ms_per_step'][k], r['gpu_ms_per_step'][k], r['host_enqueue_ms_per_step'][k]))
    assert all(h <= w * 1.05 + 0.05 for h, w in zip(r['host_enqueue_ms_per_step'], r['wall_ms_per_step']))


@pytest.mark.gpu
def test_strong_scaling_eight_ranks_against_the_whole_job_on_one_gpu():
    """`bench.py --scaling strong`: the JOB is fixed at the 8-rank shape (c2: BASELINE configs[3], 8 views x 2048 px = 16384 rays) and its views are sharded --
    `--gpus 8` (one view per rank; here the eight ranks share the test box's GPU over gloo) against `--gpus 1` (all 16384 rays on one GPU, one process): same
    rays in total, the same hits over all ranks (hit masks do not depend on the sharding), and the norm of the rank-averaged gradient equal to the single-process
    gradient's up to the ranks' own eikonal / min-sdf draws (the exact equality on shared draws: tests/test_gpu_dp.py[c4x8]).  north_star's ">= 6x at 8 GPUs
    vs 1" is value(--gpus 8 --scaling strong) / value(--gpus 1 --scaling strong)."""
    args = ['--scaling', 'strong', '--steps', '2', '--warmup', '1', '--no-cpu-baseline']
    rc, lines, err = _run(['--gpus', '8'] + args, {'MVSDF_DIST_BACKEND': 'gloo', 'OMP_NUM_THREADS': '2'}, timeout=1200)
    assert rc == 0, err
    d8 = json.loads(lines[-1])
    rc, lines, err = _run(['--gpus', '1'] + args, {}, timeout=600)
    assert rc == 0, err
    d1 = json.loads(lines[-1])
    for d, n in ((d8, 8), (d1, 1)):
        assert d['scaling'] == 'strong' and d['n_gpus'] == n and d['config']['rays_total'] == 16384 and d['config']['rays_per_gpu'] == 16384 // n, d['config']
        assert abs(d['value'] - 16384 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert d8['config']['hits_total'] == d1['config']['hits_total'] > 4000
    g8, g1 = d8['config']['grad_norm_after_all_reduce'], d1['config']['grad_norm_after_all_reduce']
    print('strong scaling on one GPU: 8 ranks %.2f ms, 1 rank %.2f ms per 16384-ray step; hits %d; |grad| %.5g vs %.5g' % (
        d8['ms_per_step'], d1['ms_per_step'], d1['config']['hits_total'], g8, g1))
    assert abs(g8 - g1) <= 0.05 * g1


@pytest.mark.gpu
def test_shipped_workload_line():
    """`--workload shipped` = the reference's real default step on one GPU: 8 views x 4096 px = 32768 rays (README.md:38, mvsdf_dtu.conf:4), 8x512 / 4x512 networks
    (mvsdf_dtu.conf:24,35), 2 source views (scene_dataset.py:104): a secondary line, same contract."""
    rc, lines, err = _run(['--workload', 'shipped', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], {}, timeout=900)
    assert rc == 0, err
    d = json.loads(lines[-1])
    c = d['config']
    assert c['rays_total'] == 32768 and c['sdf_width'] == 512 and c['src_views'] == 2 and c['views_total'] == 8 and d['dtype'] == 'f32x3'
    assert d['value'] > 0 and 0 < d['roofline']['frac'] < 1 and c['hits_total'] > 8000
    print('shipped workload (32768 rays, 8x512): %.2f ms per step = %.3g rays/s' % (d['ms_per_step'], d['value']))


@pytest.mark.gpu
def test_eight_ranks_on_one_gpu_c5_share_bf16x2_against_the_single_process_shard():
    """BASELINE configs[4] as the 8-rank run it is: --workload c5share --dtype bf16x2 --gpus 8 (32768 rays in total, bf16 weights in the tracing MLP).
    Rank 0's tracer results must not depend on the company it keeps: its hit count and its evaluated tracer rows equal those of ONE process running
    rank 0's shard alone (same views, same pixels, same seed)."""
    args = ['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--workload', 'c5share', '--dtype', 'bf16x2']
    rc, lines, err = _run(['--gpus', '8'] + args, {'MVSDF_DIST_BACKEND': 'gloo', 'OMP_NUM_THREADS': '2'}, timeout=1200)
    assert rc == 0, err
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 8 and d['config']['rays_total'] == 32768 and d['dtype'] == 'bf16x2' and d['value'] > 0
    assert d['collective_ms']['grad_all_reduce'] > 0
    assert d['ranks']['devices'] == 1 and d['ranks']['ms_min'] <= d['ranks']['ms_max']
    rc1, lines1, err1 = _run(['--shard', '0/8'] + args, {})
    assert rc1 == 0, err1
    d1 = json.loads(lines1[-1])
    s8, s1 = d['roofline']['step'], d1['roofline']['step']
    print('rank 0 of 8: %d hits, %d tracer rows; its shard alone: %d hits, %d tracer rows' % (s8['N_hit'], s8['T_trace_rows'], s1['N_hit'], s1['T_trace_rows']))
    assert s8['R'] == s1['R'] == 4096
    assert s8['N_hit'] == s1['N_hit'] and s8['T_trace_rows'] == s1['T_trace_rows'] and s8['T_reference_rows'] == s1['T_reference_rows']


@pytest.mark.gpu
def test_one_rank_under_the_launcher_runs_the_collectives_through_rccl():
    """torch.distributed.run with ONE rank and the default backend (`nccl` = RCCL on ROCm): the process group is created on the GPU and the
    step's gradient all-reduce, barrier and the MAX-of-times all-reduce all go through RCCL (identity at world size 1) -- the same code path
    the 8-GPU run takes, exercised on the one GPU of the test box."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MVSDF_DIST_BACKEND'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1', '--master-port',
           str(29600 + os.getpid() % 300), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '2', '--no-cpu-baseline']
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['value'] > 0
    assert d['collective_ms']['backend'] == 'nccl' and d['collective_ms']['grad_all_reduce'] > 0     # (world 1: no count all-reduce)


def _reference_kernel_ms():
    """The fp32 tracing MLP alone on 65 536 rows (0.64 ms on an idle MI355X): the yardstick of the gate below, timed in the same run on the same box."""
    import torch
    from helpers import sdf_packed_net
    from mvsdf_amd import ops
    from mvsdf_amd.utils import synth
    net = sdf_packed_net(synth.make_state_dict(256, 0))
    x = torch.rand(65536, 3, device='cuda') * 2 - 1
    for _ in range(3):
        ops.sdf_col0(net, x, mt=2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.sdf_col0(net, x, mt=2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


@pytest.mark.gpu
@pytest.mark.perf
@pytest.mark.parametrize('args,ratio_max', [
    ([], 4.1),                                                          # measured 2.0 ms / 0.64 ms = 3.1 (profiles/r03_bench_line.json)
    (['--workload', 'c5share', '--dtype', 'bf16x2'], 3.4),              # measured 1.6 ms / 0.64 ms = 2.5
], ids=['c2_f32', 'c5share_bf16x2'])
def test_regression_gate_of_the_headline_numbers(args, ratio_max):
    """Not a benchmark: a gate ~30 % above the committed numbers, so that a change which silently loses a fusion, the native step driver or an engine
    path fails the suite instead of surfacing as a slower bench line a round later.  RELATIVE: the step time is compared with a micro-kernel (the fp32
    tracing MLP alone) timed in this same run, so a throttled or shared GPU moves both (the absolute form of round 3 could fail unrelated changes)."""
    t_ref = _reference_kernel_ms()
    rc, lines, err = _run(args + ['--steps', '40', '--warmup', '15', '--no-cpu-baseline'], {})
    assert rc == 0, err
    d = json.loads(lines[-1])
    print('%s: %.3f ms/step = %.2f x the reference kernel (%.3f ms), roofline %s' % (' '.join(args) or 'c2', d['ms_per_step'], d['ms_per_step'] / t_ref, t_ref,
                                                                                     d['roofline'].get('frac')))
    assert d['ms_per_step'] < ratio_max * t_ref, (d['ms_per_step'], t_ref)
