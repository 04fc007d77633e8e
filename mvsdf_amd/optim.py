"""Optimiser tail of the training step on flat buffers.

Reference (code/training/idr_train.py:113,289-302): torch.optim.Adam(model.parameters(), lr) preceded by an all-parameter gradient
norm and, once train_progress >= phase[0], torch.nn.utils.clip_grad_norm_(model.parameters(), grad_cap).  Here the parameters,
gradients and both Adam moments each live in ONE flat fp32 buffer (every Parameter / .grad / state tensor is a view into it), so the
whole tail is two HIP launches (mvsdf_adam_step) and the data-parallel all-reduce needs no packing.

`FlatAdam` is a torch.optim.Optimizer: lr schedulers (MultiStepLR, idr_train.py:119) drive `param_groups[0]['lr']`, and
state_dict() / load_state_dict() use torch.optim.Adam's layout (per-parameter 'step', 'exp_avg', 'exp_avg_sq'), so the reference's
OptimizerParameters/*.pth checkpoints (idr_train.py:171-177) load and save unchanged."""
import ctypes as C

import torch

from ._lib import lib, check
from .parallel import all_reduce_sum_


try:                                                             # torch's global optimizer step hooks (private names: absent / renamed in another build = no global hooks)
    from torch.optim.optimizer import _global_optimizer_post_hooks as _gpost, _global_optimizer_pre_hooks as _gpre
    _GLOBAL_HOOKS = (_gpre, _gpost)
except ImportError:
    _GLOBAL_HOOKS = ({}, {})


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        params = [p for p in params if p.requires_grad]
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError('FlatAdam implements the reference\'s optimiser: Adam(lr) without weight decay / amsgrad (idr_train.py:113)')
        # weight_decay / amsgrad are carried in the group only so that state_dict() is a COMPLETE torch.optim.Adam group: the reference's
        # Adam can load a file written here and step (its step() reads both keys)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        assert len(self.param_groups) == 1, 'FlatAdam keeps one parameter group (the reference has one, idr_train.py:113)'
        ps = self.param_groups[0]['params']
        p0 = ps[0]
        assert all(p.is_cuda and p.dtype == torch.float32 and p.device == p0.device for p in ps), 'fp32 parameters on one GPU expected'
        total = sum(p.numel() for p in ps)
        self.flat_p = torch.empty(total, dtype=torch.float32, device=p0.device)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self._ws = torch.empty(lib().mvsdf_adam_ws_floats(), dtype=torch.float32, device=p0.device)
        self.norm_and_coef = torch.zeros(2, dtype=torch.float32, device=p0.device)     # {||grad||, clip coefficient} of the last step
        self._t = 0
        self._grad_scale = 1.0                                                         # 1 / world between all_reduce_mean() and step()
        self._zero_version = None                                                      # flat_g's version counter right after a step(zero_grad=True) left zeros in it
        self._slices = []
        off = 0
        for p in ps:
            n = p.numel()
            self._slices.append((off, n))
            self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + n].view(p.shape)                             # the Parameter now lives in the flat buffer
            off += n
        self._attach_grads()
        self._attach_state()

    # ---- views
    def _attach_grads(self):
        for p, (off, n) in zip(self.param_groups[0]['params'], self._slices):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + n].view(p.shape)
            p._mv_grad_sink = True                                                      # functional._FoldNet adds into p.grad directly
        self._grad_views = [p.grad for p in self.param_groups[0]['params']]            # identity check in _sync_grads

    def _attach_state(self):
        for p, (off, n) in zip(self.param_groups[0]['params'], self._slices):
            self.state[p] = {'step': torch.tensor(float(self._t)), 'exp_avg': self.flat_m[off:off + n].view(p.shape),
                             'exp_avg_sq': self.flat_v[off:off + n].view(p.shape)}

    # ---- the step
    def zero_grad(self, set_to_none=False):
        """One memset of the flat gradient buffer; gradients stay attached (views) whatever `set_to_none` says.  No launch at all when the last
        step(zero_grad=True) left zeros there and nothing wrote since (every write -- autograd's accumulation, the gradient sink, a collective through torch --
        moves the buffer's version counter, which the views share)."""
        if self._zero_version is None or self.flat_g._version != self._zero_version:
            self.flat_g.zero_()
        self._zero_version = None
        if not all(p.grad is v for p, v in zip(self.param_groups[0]['params'], self._grad_views)):
            self._attach_grads()

    def backward(self, loss):
        """loss.backward() with the direct gradient sink on (functional.grad_sink): the weight_norm-fold backward adds dv / dg / db of
        both networks into the flat gradient buffer in ONE launch instead of handing ~40 tensors to autograd's AccumulateGrad."""
        from .functional import grad_sink
        from .native_step import direct_backward
        with grad_sink():
            if not direct_backward(loss):                        # the native step's own two-node graph: run in line, no engine hand-over
                loss.backward()

    def _sync_grads(self):
        """Every p.grad must still be its view of flat_g when the flat kernels read it.  `model.zero_grad()` (set_to_none=True by default)
        or `p.grad = None` detach them: autograd then allocates fresh .grad tensors.  Copy such strays in and re-attach.  A parameter
        WITHOUT a gradient counts as a zero gradient, i.e. it still moves by its momentum -- what the reference's pinned torch 1.7.1
        does (its zero_grad() zero-fills, idr_train.py:283), whereas torch >= 2.0's Adam would skip a None gradient."""
        views = self._grad_views
        if all(p.grad is v for p, v in zip(self.param_groups[0]['params'], views)):     # the common case: nothing was detached
            return
        base = self.flat_g.data_ptr()
        for i, (p, (off, n)) in enumerate(zip(self.param_groups[0]['params'], self._slices)):
            if p.grad is None:
                self.flat_g[off:off + n].zero_()
            elif p.grad.data_ptr() != base + 4 * off:
                self.flat_g[off:off + n].copy_(p.grad.detach().reshape(-1))
            else:
                views[i] = p.grad
                continue
            p.grad = self.flat_g[off:off + n].view(p.shape)
            views[i] = p.grad

    def all_reduce_mean(self, defer_scale=False):
        """The step's one gradient collective (RCCL over xGMI): SUM all-reduce of the flat gradient buffer, then the rank average.
        Default: every p.grad holds the MEAN when this returns (one scaling pass over the 3 MB buffer), so a loop ported from the reference
        may read or clip the gradients between the collective and step() (idr_train.py:289-294).
        defer_scale=True: the buffer keeps the SUM and the 1 / world is applied inside step()'s Adam launch (no separate pass; what bench.py
        uses) -- until step() runs, p.grad is world x too large; grad_norm() / step(grad_cap=...) account for it, external code must not
        read the gradients in between.  No-op without a process group."""
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return                                                                     # no process group: nothing to reduce (step() checks the gradient views itself)
        self._sync_grads()
        scale = all_reduce_sum_(self.flat_g)
        if defer_scale:
            self._grad_scale = scale
        elif scale != 1.0:
            self.flat_g.mul_(scale)

    def step(self, closure=None, grad_cap=None, zero_grad=False):
        """grad-norm + optional clip_grad_norm_(grad_cap) + Adam in two launches.  `norm_and_coef` holds the norm afterwards.
        zero_grad=True: the update pass leaves ZEROS in the gradients instead of the scaled / clipped values (torch's Adam leaves them; the reference's loop
        zeroes them at the top of the next iteration, idr_train.py:283): that zero_grad() then costs no launch.  Do not use it when the gradients are read
        after step().
        torch.optim.Optimizer's step pre / post hooks run if any are registered (the profiler range torch wraps around step() is skipped:
        it costs more host time than the two launches)."""
        assert closure is None
        _global_optimizer_pre_hooks, _global_optimizer_post_hooks = _GLOBAL_HOOKS
        hooks = bool(self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _global_optimizer_pre_hooks or _global_optimizer_post_hooks)
        args, kwargs = (self,), {'closure': closure, 'grad_cap': grad_cap, 'zero_grad': zero_grad}
        if hooks:
            for h in list(_global_optimizer_pre_hooks.values()) + list(self._optimizer_step_pre_hooks.values()):
                r = h(self, args, kwargs)
                if r is not None:
                    args, kwargs = r
                    grad_cap = kwargs.get('grad_cap', grad_cap)
                    zero_grad = kwargs.get('zero_grad', zero_grad)
        g = self.param_groups[0]
        self._sync_grads()
        self._t += 1
        fp = self.flat_p
        check(lib().mvsdf_adam_step_fused(fp.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(),
                                          fp.numel(), float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                          self._t, float(grad_cap) if grad_cap else 0.0, float(self._grad_scale), 1 if zero_grad else 0,
                                          self.norm_and_coef.data_ptr(), self._ws.data_ptr(),
                                          C.c_void_p(torch._C._cuda_getCurrentRawStream(fp.device.index if fp.device.index is not None else torch.cuda.current_device()))),
              'mvsdf_adam_step_fused')
        self._grad_scale = 1.0
        # (the launch wrote the gradient buffer through a raw pointer: move its version counter, then remember it when zeros were left)
        torch.autograd.graph.increment_version(self.flat_g)
        self._zero_version = self.flat_g._version if zero_grad else None
        # the launch wrote the parameters through a raw pointer: tell autograd (an in-place update), so that a backward whose forward ran BEFORE this step
        # raises like it does after torch.optim.Adam.step() instead of mixing old activations with the new weights (native_step._NativeStepFn.backward and
        # autograd's own saved-tensor check compare these version counters)
        torch.autograd.graph.increment_version(g['params'])
        if hooks:
            for h in list(self._optimizer_step_post_hooks.values()) + list(_global_optimizer_post_hooks.values()):
                h(self, args, kwargs)
    step.hooked = True                                       # torch.optim.Optimizer would otherwise wrap it in its profiler hook

    def grad_norm(self):
        return self.norm_and_coef[0]

    # ---- torch.optim.Adam checkpoint layout
    def state_dict(self):
        for st in self.state.values():
            st['step'] = torch.tensor(float(self._t))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        for grp in state_dict.get('param_groups', []):
            if grp.get('weight_decay', 0) != 0 or grp.get('amsgrad', False):
                raise NotImplementedError('FlatAdam: checkpoint of an Adam with weight decay / amsgrad')
        super().load_state_dict(state_dict)                  # leaves fresh (non-view) state tensors behind: copy them into the flat buffers
        ps = self.param_groups[0]['params']
        steps = set()
        for p, (off, n) in zip(ps, self._slices):
            st = self.state.get(p)
            if st:
                self.flat_m[off:off + n].copy_(st['exp_avg'].reshape(-1))
                self.flat_v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                steps.add(int(float(st['step'])))
        assert len(steps) <= 1, 'per-parameter step counts differ'
        self._t = steps.pop() if steps else 0
        self._attach_state()
