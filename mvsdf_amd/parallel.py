"""Data-parallel ray sharding: one process per GPU, ONE all-reduce (RCCL over xGMI; gloo in the CPU tests) on a flat
fp32 gradient bucket per step.  The reference is single-GPU (exp_runner.py:24-28); this layer is new.

Rays are independent; the only coupling is the parameter gradient.  Views are sharded across ranks (B views -> B/world each);
`.grad` of every parameter is a VIEW into one flat buffer, so backward accumulates straight into the bucket and the
collective needs no packing.  Loss normalisation: gradients are AVERAGED over ranks (mean of per-rank losses), the DDP convention;
that equals the single-process loss for the per-view / per-ray means, and IDRLoss divides the count-normalised terms (eikonal / depth /
surface BCE) by the counts summed over the ranks, so the averaged gradient is the single-process one (tests/test_gpu_dp.py)."""
import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_reduce_sum_(flat):
    """The step's one gradient collective: SUM all-reduce of a flat gradient buffer (RCCL over xGMI with the `nccl` backend; gloo in the
    CPU tests).  No-op without an initialised process group.  -> the factor that turns the sum into the rank average (1 / world): the
    caller folds it into its next pass over the buffer (optim.FlatAdam: inside the Adam launch) instead of a separate division."""
    w = world_size()
    if dist.is_available() and dist.is_initialized():              # also at world size 1 (the collective then is RCCL's identity)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return 1.0 / w


def all_reduce_mean_(flat):
    """SUM all-reduce + division by the world size (for callers without a later pass to fold the factor into)."""
    scale = all_reduce_sum_(flat)
    if scale != 1.0:
        flat.mul_(scale)
    return flat


class FlatGradBucket:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)          # autograd accumulates in place into the bucket
            p._mv_grad_sink = True                              # functional._FoldNet may add into p.grad directly
            off += n

    def zero(self):
        self.flat.zero_()
        off = 0
        for p in self.params:                                   # re-attach (an optimizer's zero_grad(set_to_none) would detach)
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat[off:off + n].data_ptr():
                p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def backward(self, loss):
        """loss.backward() with the direct gradient sink on (functional.grad_sink): gradients are added into the bucket by one launch."""
        from .functional import grad_sink
        with grad_sink():
            loss.backward()

    def sync_grads(self):
        """Make the bucket hold every parameter's gradient again after something detached `.grad` from it (model.zero_grad(), whose
        default set_to_none=True makes autograd allocate fresh tensors; `p.grad = None`): copy the stray gradient in (zeros for a
        parameter without one) and re-attach the view."""
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
                p.grad = self.flat[off:off + n].view_as(p)
            elif p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
                p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def all_reduce_mean(self):
        """One collective for all gradients; no-op without an initialised process group."""
        self.sync_grads()
        all_reduce_mean_(self.flat)

    def grad_norm(self):
        self.sync_grads()                                       # gradients detached from the bucket (model.zero_grad()) are copied in first
        return self.flat.norm()

    def clip_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ on the bucket (idr_train.py:291-294): one norm, one scale."""
        self.sync_grads()
        total = self.flat.norm()
        coef = (max_norm / (total + 1e-6)).clamp(max=1.0)
        self.flat.mul_(coef)
        return total


def shard_views(batch, rank, world):
    """Slice every [B, ...] tensor of a model-input / ground-truth dict to this rank's views (B must divide by world).
    Depth maps are NOT sliced: the depth loss carves every sample point against all B views (loss.py:39-40)."""
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() >= 1 and k not in ('depths', 'depth_cams'):
            B = v.shape[0]
            assert B % world == 0, 'views must divide evenly across ranks'
            per = B // world
            out[k] = v[rank * per:(rank + 1) * per]
        else:
            out[k] = v
    return out
