"""N > 1 path on CPU: world_size 2, gloo.  The flat gradient bucket + one all-reduce reproduces the single-process gradient
of the mean loss over the un-sharded batch; view sharding keeps depth maps whole."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvsdf_amd.parallel import FlatGradBucket, shard_views


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Softplus(beta=100), torch.nn.Linear(16, 3))


def _batch():
    g = torch.Generator().manual_seed(1)
    return {'x': torch.randn(8, 5, 6, generator=g), 'y': torch.randn(8, 5, 3, generator=g), 'depths': torch.randn(8, 1, 1, 4, 4, generator=g),
            'depth_cams': torch.randn(8, 1, 2, 4, 4, generator=g), 'scalar': 3}


def _loss(m, b):
    return ((m(b['x']) - b['y']).abs()).mean()


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    m = _model()
    bucket = FlatGradBucket(m.parameters())
    sh = shard_views(_batch(), rank, world)
    assert sh['x'].shape[0] == 8 // world and sh['depths'].shape[0] == 8 and sh['scalar'] == 3
    bucket.zero()
    _loss(m, sh).backward()
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in m.parameters())
    bucket.all_reduce_mean()
    total = bucket.clip_(1e9)
    if rank == 0:
        ret['flat'] = bucket.flat.clone()
        ret['norm'] = float(total)
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_allreduce_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    m = _model()
    _loss(m, _batch()).backward()
    ref = torch.cat([p.grad.flatten() for p in m.parameters()])
    assert torch.allclose(ret['flat'], ref, rtol=1e-5, atol=1e-7)
    assert abs(ret['norm'] - float(ref.norm())) < 1e-5


def test_bucket_single_process_clip_matches_torch():
    m, m2 = _model(), _model()
    b = _batch()
    bucket = FlatGradBucket(m.parameters())
    bucket.zero()
    (_loss(m, b) * 50).backward()
    bucket.all_reduce_mean()                      # no process group: no-op
    bucket.clip_(0.5)
    (_loss(m2, b) * 50).backward()
    torch.nn.utils.clip_grad_norm_(m2.parameters(), 0.5)
    for p, q in zip(m.parameters(), m2.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-8)
    bucket.zero()
    assert float(bucket.flat.abs().sum()) == 0 and all(p.grad is not None for p in m.parameters())
